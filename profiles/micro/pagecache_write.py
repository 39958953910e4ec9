"""How fast does this box's page cache take buffered writes -- one file vs several, pwrite vs a shared mapping?
(the D2H + write phase of `sufr create` writes 12-15 GB into ONE file; python profiles/micro/pagecache_write.py [GiB] [dir])"""
import mmap, os, sys, threading, time
G = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
D = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("TMPDIR", "/tmp")
PIECE = 32 << 20
N = int(G * (1 << 30)) // PIECE
buf = bytearray(os.urandom(1 << 20) * 32)
print("dir", D, "fs:", os.popen(f"df -T {D} | tail -1").read().strip())

def run(label, threads, files, use_mmap=False, prefault=False):
    paths = [f"{D}/pcw_{i}.bin" for i in range(files)]
    fds = [os.open(p, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644) for p in paths]
    per = N // files
    maps = []
    t0 = time.perf_counter()
    if use_mmap:
        for fd in fds:
            os.ftruncate(fd, per * PIECE)
            maps.append(mmap.mmap(fd, per * PIECE, mmap.MAP_SHARED, mmap.PROT_WRITE | mmap.PROT_READ))
    nxt = [0]; lock = threading.Lock()
    def work():
        while True:
            with lock:
                i = nxt[0]; nxt[0] += 1
            if i >= per * files: return
            f, k = i % files, i // files
            if use_mmap:
                if prefault and hasattr(mmap, "MADV_POPULATE_WRITE"):
                    maps[f].madvise(mmap.MADV_POPULATE_WRITE, k * PIECE, PIECE)
                maps[f][k * PIECE:(k + 1) * PIECE] = buf
            else:
                os.pwrite(fds[f], buf, k * PIECE)
    th = [threading.Thread(target=work) for _ in range(threads)]
    for x in th: x.start()
    for x in th: x.join()
    for m in maps: m.close()
    for fd in fds: os.close(fd)
    dt = time.perf_counter() - t0
    for p in paths: os.unlink(p)
    print(f"{label:44s} {per * files * PIECE / dt / 1e9:6.2f} GB/s")

run("pwrite, 1 thread, 1 file", 1, 1)
run("pwrite, 8 threads, 1 file", 8, 1)
run("pwrite, 8 threads, 8 files", 8, 8)
run("pwrite, 16 threads, 16 files", 16, 16)
run("mmap, 8 threads, 1 file", 8, 1, True)
run("mmap + MADV_POPULATE_WRITE, 8 threads, 1 file", 8, 1, True, True)
run("mmap, 16 threads, 1 file", 16, 1, True)
