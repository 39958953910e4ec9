import os, sys, threading, time
def run(path, threads, total=12 << 30, piece=32 << 20):
    buf = bytes(piece)
    fd = os.open(path, os.O_CREAT | os.O_WRONLY | os.O_TRUNC, 0o644)
    n = total // piece
    nxt = [0]; lock = threading.Lock()
    def w():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= n: return
            os.pwrite(fd, buf, i * piece)
    t0 = time.perf_counter()
    th = [threading.Thread(target=w) for _ in range(threads)]
    [t.start() for t in th]; [t.join() for t in th]
    os.close(fd)
    dt = time.perf_counter() - t0
    os.unlink(path)
    return total / dt / 1e9
for d in ("/tmp", "/dev/shm", os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"):
    for t in (1, 4, 12, 32):
        try:
            print(d, t, "threads: %.1f GB/s" % run(os.path.join(d, "wr_test.bin"), t), flush=True)
        except Exception as e:
            print(d, t, "failed", e)
