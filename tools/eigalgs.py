"""rocSOLVER symmetric eigensolver variants on KFAC-factor-like matrices: time and accuracy of syevd / syevdj / syevj.
    python tools/eigalgs.py [n ...]"""
import ctypes, sys, time
import torch
rs = ctypes.CDLL('/opt/rocm/lib/librocsolver.so')
rb = ctypes.CDLL('/opt/rocm/lib/librocblas.so')
c_p, c_i, c_f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
h = c_p()
assert rb.rocblas_create_handle(ctypes.byref(h)) == 0
rb.rocblas_set_stream(h, c_p(torch.cuda.current_stream().cuda_stream))
EV, UP, ASC = 211, 121, 252
dev = 'cuda'
ns = [int(x) for x in sys.argv[1:]] or [257, 513, 1569]
for n in ns:
    g = torch.Generator(device='cpu').manual_seed(n)
    X = torch.randn(640, n, generator=g)                      # rank-deficient for n > 640, like the FC factor
    A0 = (X.t() @ X / 640).to(dev)
    A0 = 0.95 * A0 + 0.05 * torch.eye(n, device=dev) * A0.diagonal().mean()
    ref_w = torch.linalg.eigvalsh(A0.double().cpu())
    def check(Q, W):
        # rows of Q (column-major eigenvectors) -> A ~ Q^T diag(W) Q
        R = (Q.t() * W) @ Q
        return float((R - A0).norm() / A0.norm()), float((W.cpu().double().sort().values - ref_w).abs().max() / ref_w.abs().max())
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    W = torch.empty(n, device=dev); E = torch.empty(n, device=dev)
    res = torch.zeros(1, device=dev); nsw = torch.zeros(1, dtype=torch.int32, device=dev)
    algs = {
        'syevd': lambda A: rs.rocsolver_ssyevd(h, EV, UP, n, c_p(A.data_ptr()), n, c_p(W.data_ptr()), c_p(E.data_ptr()), c_p(info.data_ptr())),
        'syevdj': lambda A: rs.rocsolver_ssyevdj(h, EV, UP, n, c_p(A.data_ptr()), n, c_p(W.data_ptr()), c_p(info.data_ptr())),
        'syevj': lambda A: rs.rocsolver_ssyevj(h, ASC, EV, UP, n, c_p(A.data_ptr()), n, c_f(1e-7), c_p(res.data_ptr()), 30, c_p(nsw.data_ptr()), c_p(W.data_ptr()), c_p(info.data_ptr())),
    }
    for name, fn in algs.items():
        ts = []
        for rep in range(3):
            A = A0.clone()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            st = fn(A)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        rec, werr = check(A, W)
        print('n=%5d %-7s status %d info %d  %.2f ms (first %.1f)  recon err %.2e  eigenvalue err %.2e  %s' % (
            n, name, st, int(info.item()), min(ts[1:]), ts[0], rec, werr, ('sweeps %d' % int(nsw.item())) if name == 'syevj' else ''), flush=True)
