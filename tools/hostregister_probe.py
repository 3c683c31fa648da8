# probe: can the DMA engines read a memory-mapped (page-cache) file directly?  hipHostRegister on the mapping + hipMemcpyAsync
import ctypes, mmap, os, time, numpy as np, torch
hip = ctypes.CDLL("libamdhip64.so")
path = "/tmp/hr_probe.bin"
size = 1 << 30
if not os.path.exists(path):
    np.random.default_rng(0).integers(0, 255, size, dtype=np.uint8).tofile(path)
fd = os.open(path, os.O_RDONLY)
dst = torch.empty(size, dtype=torch.uint8, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
for prot, flags, name in ((mmap.PROT_READ, mmap.MAP_SHARED, "shared ro"), (mmap.PROT_READ | mmap.PROT_WRITE, mmap.MAP_PRIVATE, "private rw")):
    mm = mmap.mmap(fd, size, flags=flags, prot=prot)
    addr = ctypes.addressof(ctypes.c_char.from_buffer(mm)) if prot & mmap.PROT_WRITE else np.frombuffer(mm, dtype=np.uint8).ctypes.data
    for reg_flag in (0, 8):
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(ctypes.c_void_p(addr), ctypes.c_size_t(size), ctypes.c_uint(reg_flag))
        t1 = time.perf_counter()
        print(name, "flag", reg_flag, "hipHostRegister rc", rc, "seconds", round(t1 - t0, 3))
        if rc == 0:
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            rc2 = hip.hipMemcpyAsync(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(addr), ctypes.c_size_t(size), 1, ctypes.c_void_p(stream))
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            ok = bool((dst[:4096].cpu().numpy() == np.frombuffer(mm, dtype=np.uint8, count=4096)).all())
            print("   memcpy rc", rc2, "GB/s", round(size / (t3 - t2) / 1e9, 1), "data ok", ok)
            t4 = time.perf_counter(); hip.hipHostUnregister(ctypes.c_void_p(addr)); print("   unregister", round(time.perf_counter() - t4, 3))
            break
# baseline: pageable memcpy straight from the mapping (the runtime stages it itself)
mm = mmap.mmap(fd, size, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
addr = np.frombuffer(mm, dtype=np.uint8).ctypes.data
torch.cuda.synchronize(); t0 = time.perf_counter()
rc = hip.hipMemcpy(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(addr), ctypes.c_size_t(size), 1)
torch.cuda.synchronize(); print("plain hipMemcpy from the mapping rc", rc, "GB/s", round(size / (time.perf_counter() - t0) / 1e9, 1))
# pieces: page-sized (1 MB) and 8 MB pageable copies from the mapping, one thread and four
import threading
for piece in (1 << 20, 8 << 20, 64 << 20):
    for nthr in (1, 4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        def work(t):
            hip.hipSetDevice(0)
            for off in range(t * piece, size, nthr * piece):
                hip.hipMemcpy(ctypes.c_void_p(dst.data_ptr() + off), ctypes.c_void_p(addr + off), ctypes.c_size_t(piece), 1)
        ths = [threading.Thread(target=work, args=(t,)) for t in range(nthr)]
        [t.start() for t in ths]; [t.join() for t in ths]
        torch.cuda.synchronize(); print("pageable pieces of", piece >> 20, "MB, threads", nthr, "GB/s", round(size / (time.perf_counter() - t0) / 1e9, 1))
