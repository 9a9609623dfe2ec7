"""A/B of emp_conv_bn_act_nhwc between two builds of the library on the same device (box-to-box clock differences are
larger than the effect looked for).  `python tools/ab_conv.py <libA.so> <libB.so>`"""
import ctypes
import sys

import torch

P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
ARGS = [P, P, P, P, P, L, I] + [I] * 10 + [P, L, P]
#        name                      cin   cout  hw  k  pad res
CASES = [('head.pw 256->256 @256', 256, 256, 256, 1, 0, False), ('l4.down 1024->2048 @64', 1024, 2048, 64, 1, 0, False),
         ('l4.conv3 512->2048 @64 +res', 512, 2048, 64, 1, 0, True), ('l1.conv3 64->256 @256 +res', 64, 256, 256, 1, 0, True),
         ('l3.conv2 256->256 3x3 @128 s1', 256, 256, 128, 3, 1, False), ('l2.conv1 512->128 @128', 512, 128, 128, 1, 0, False)]


def main():
    libs = [ctypes.CDLL(p) for p in sys.argv[1:3]]
    for lib in libs:
        lib.emp_conv_bn_act_nhwc.restype = I
        lib.emp_conv_bn_act_nhwc.argtypes = ARGS
    B = 32
    st = torch.cuda.current_stream().cuda_stream
    for name, cin, cout, hw, k, pad, res in CASES:
        x = torch.randn(B, hw, hw, cin, device='cuda')
        w = torch.randn(cout, k, k, cin, device='cuda') * 0.02
        sc, sh = torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda')
        r = torch.randn(B, hw, hw, cout, device='cuda') if res else None
        out = torch.empty(B, hw, hw, cout, device='cuda')
        ms = []
        outs = []
        for rep in range(3):                                # interleaved: A B A B A B
            for lib in libs:
                def go():
                    rc = lib.emp_conv_bn_act_nhwc(x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                                  r.data_ptr() if res else None, 0, 1, B, hw, hw, cin, cout, k, k, 1, pad, 1,
                                                  out.data_ptr(), 0, st)
                    assert rc == 0
                for _ in range(3):
                    go()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    go()
                e1.record()
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1) / 20)
                if rep == 0:
                    outs.append(out.clone())
        a, b = min(ms[0::2]), min(ms[1::2])
        print(f'{name:32s} A {a:7.4f} ms  B {b:7.4f} ms  B/A {b / a:6.3f}  identical {bool(torch.equal(outs[0], outs[1]))}', flush=True)


if __name__ == '__main__':
    main()
