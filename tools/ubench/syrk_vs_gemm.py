import ctypes as C, torch, time
blas = C.CDLL("librocblas.so")
h = C.c_void_p(); assert blas.rocblas_create_handle(C.byref(h)) == 0
blas.rocblas_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
one = C.c_float(1.0); zero = C.c_float(0.0)
for n, m, P in ((128, 4096, 512), (64, 2000, 2048), (256, 8192, 128)):
    J = torch.rand(P, m, n, device="cuda") - 0.5
    H = torch.empty(P, n, n, device="cuda"); H2 = torch.zeros(P, n, n, device="cuda")
    def gemm():
        return blas.rocblas_sgemm_strided_batched(h, 111, 112, n, n, m, C.byref(one), C.c_void_p(J.data_ptr()), n, C.c_int64(m*n), C.c_void_p(J.data_ptr()), n, C.c_int64(m*n), C.byref(zero), C.c_void_p(H.data_ptr()), n, C.c_int64(n*n), P)
    def syrk():  # C = A A^T, A = n x m col-major (ld n), upper
        return blas.rocblas_ssyrk_strided_batched(h, 121, 111, n, m, C.byref(one), C.c_void_p(J.data_ptr()), n, C.c_int64(m*n), C.byref(zero), C.c_void_p(H2.data_ptr()), n, C.c_int64(n*n), P)
    for name, f in (("gemm", gemm), ("syrk", syrk)):
        assert f() == 0; torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        print(n, m, P, name, e0.elapsed_time(e1) / 5, "ms")
    ref = torch.bmm(J.transpose(1, 2), J)
    print("gemm err", float((H - ref).abs().max()), "syrk upper err", float((torch.triu(H2.transpose(1,2)) - torch.triu(ref.transpose(1,2))).abs().max()), float((torch.tril(H2.transpose(1,2)) - torch.tril(ref.transpose(1,2))).abs().max()))
