"""Does a hipMemsetAsync captured into a hipGraph clear the words on every replay?"""
import ctypes, torch
rt = ctypes.CDLL("libamdhip64.so")
rt.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for nbytes in (8, 64, 4096):
    buf = torch.full((nbytes // 4,), 5, dtype=torch.int32, device="cuda")
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        rc = rt.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        buf.add_(1)
    torch.cuda.synchronize()
    print(nbytes, "rc", rc, "after capture", buf[:2].tolist())
    for i in range(4):
        g.replay()
        torch.cuda.synchronize()
        print(nbytes, "replay", i, buf[:2].tolist(), buf[-1].item())
