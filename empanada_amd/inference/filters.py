"""Tracker filters, reference names (``empanada/inference/filters.py``:9-43)."""

__all__ = ['remove_small_objects', 'remove_pancakes']


def remove_small_objects(object_tracker, min_size=64):
    """filters.py:9-24 -- delete instances with fewer than min_size voxels, in place."""
    for instance_id in list(object_tracker.instances.keys()):
        if object_tracker.instances[instance_id]['runs'].sum() < min_size:
            del object_tracker.instances[instance_id]


def remove_pancakes(object_tracker, min_span=4):
    """filters.py:26-43 -- delete instances whose bounding box is thinner than min_span on any axis."""
    for instance_id in list(object_tracker.instances.keys()):
        box = object_tracker.instances[instance_id]['box']
        if any(span < min_span for span in (box[3] - box[0], box[4] - box[1], box[5] - box[2])):
            del object_tracker.instances[instance_id]
