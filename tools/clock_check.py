import sys, time, ctypes as C, subprocess, threading
sys.path.insert(0, "/root/repo")
import torch, micro_aes_amd as uaes
L = uaes.engine()
key, ctr0 = bytes(range(16)), bytes(range(12)) + b"\0\0\0\1"
n = 1 << 30
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
st = torch.cuda.current_stream(); side = torch.cuda.Stream()
out = torch.zeros(2, dtype=torch.int64, device="cuda")
smi = []
def poll():
    for _ in range(10):
        r = subprocess.run("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr -s ' ' | tr '\\n' ';'", shell=True, capture_output=True, text=True)
        smi.append(r.stdout.strip()); time.sleep(0.4)
def load(seconds):
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(8): uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=n, stream=st)
        st.synchronize()
# idle probe
L.uaes_clock_probe_dev(C.c_void_p(out.data_ptr()), 20000, C.c_void_p(side.cuda_stream)); torch.cuda.synchronize()
c, t = out.tolist(); print("idle probe: %.0f MHz" % (c / (t / 100.0)))
th = threading.Thread(target=poll); th.start()
load(1.0)
for i in range(5):
    L.uaes_clock_probe_dev(C.c_void_p(out.data_ptr()), 20000, C.c_void_p(side.cuda_stream))
    load(0.3); torch.cuda.synchronize()
    c, t = out.tolist(); print("probe under CTR load: %.0f MHz (%d ticks)" % (c / (t / 100.0), t))
# back to back, no host synchronisation inside the window (what bench.py's timed steps look like)
for i in range(3):
    L.uaes_clock_probe_dev(C.c_void_p(out.data_ptr()), 20000, C.c_void_p(side.cuda_stream))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(60): uaes.ctr_xcrypt_dev(key, ctr0, 0, src, dst, nbytes=n, stream=st)
    e1.record(st)
    torch.cuda.synchronize()
    c, t = out.tolist(); ms = e0.elapsed_time(e1) / 60
    mhz = c / (t / 100.0)
    print("probe, 60 steps back to back: %.0f MHz; %.4f ms per GiB = %.1f GiB/s = %.2f clk per block per CU" % (mhz, ms, 1 / ms * 1e3 / 1.0, 256 * mhz * 1e6 * ms * 1e-3 / 2**26))
th.join()
for s in smi: print("rocm-smi:", s[:200])
