"""dev helper: library GEMM times for the head's shapes (P = 56 000) under the available BLAS back ends."""
import sys, time, torch
dev = torch.device("cuda:0")
P = 56000
shapes = [("conv6 fwd  [P,192]x[192,1024]", (P, 192), (1024, 192), "nt"),
          ("conv7 fwd  [P,192]x[192,512]", (P, 192), (512, 192), "nt"),
          ("conv8 fwd  [P,512]x[512,256]", (P, 512), (256, 512), "nt"),
          ("conv8 dX   [P,256]x[256,512]", (P, 256), (256, 512), "nn"),
          ("conv7 dX   [P,512]x[512,192]", (P, 512), (512, 192), "nn"),
          ("conv8 dW   [256,P]x[P,512]", (P, 256), (P, 512), "tn"),
          ("conv7 dW   [512,P]x[P,192]", (P, 512), (P, 192), "tn"),
          ("gram       [192,P]x[P,192]", (P, 192), (P, 192), "tn"),
          ("cat M      [P,192]x[192,192]", (P, 192), (192, 192), "nn")]


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for lib in ("default", "hipblaslt", "cublas"):
    if lib != "default":
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:
            print(lib, "unavailable", e)
            continue
    print("== blas:", lib, flush=True)
    for name, sa, sb, mode in shapes:
        a = torch.randn(sa, device=dev)
        b = torch.randn(sb, device=dev)
        if mode == "nt":
            fn = lambda: torch.mm(a, b.t())
            alt = lambda: torch.mm(b, a.t()).t()
            fl = 2 * sa[0] * sa[1] * sb[0]
        elif mode == "nn":
            fn = lambda: torch.mm(a, b)
            alt = lambda: torch.mm(b.t(), a.t()).t()
            fl = 2 * sa[0] * sa[1] * sb[1]
        else:
            fn = lambda: torch.mm(a.t(), b)
            alt = lambda: torch.mm(b.t(), a).t()
            fl = 2 * sa[0] * sa[1] * sb[1]
        u1, u2 = t(fn), t(alt)
        print("  %-34s %7.1f us (%5.1f TF/s)   transposed form %7.1f us" % (name, u1, fl / u1 / 1e6, u2), flush=True)
