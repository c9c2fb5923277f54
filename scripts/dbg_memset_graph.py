import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
d = torch.device("cuda:0")
for nbytes in (8, 256, 4096, 4 << 20, (4 << 20) + 12):
    buf = torch.ones(nbytes // 4 + 4, dtype=torch.float32, device=d)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        y = buf + 1
    torch.cuda.synchronize()
    res = []
    for i in range(3):
        buf.fill_(5.0); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        res.append((float(buf[0]), float(buf[nbytes // 4 - 1]), float(y[0])))
    print(nbytes, "rc", rc, res)
