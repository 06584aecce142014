import torch, time
n = 1 << 30
h_in = torch.empty(n, dtype=torch.uint8).pin_memory(); h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_in = torch.empty(n, dtype=torch.uint8, device="cuda"); d_out = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
def h2d():
    with torch.cuda.stream(s1): d_in.copy_(h_in, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): h_out.copy_(d_out, non_blocking=True)
def both(): h2d(); d2h()
def h2d_split(k=4):
    c = n // k
    for i in range(k):
        with torch.cuda.stream(ss[i]): d_in[i*c:(i+1)*c].copy_(h_in[i*c:(i+1)*c], non_blocking=True)
ss = [torch.cuda.Stream() for _ in range(4)]
print("H2D alone  %.1f GB/s" % (n / t(h2d) / 1e9))
print("D2H alone  %.1f GB/s" % (n / t(d2h) / 1e9))
print("both       %.1f GB/s each way" % (n / t(both) / 1e9))
print("H2D split over 4 streams %.1f GB/s" % (n / t(h2d_split) / 1e9))
def both_split(k=2):
    c = n // k
    for i in range(k):
        with torch.cuda.stream(ss[i]): d_in[i*c:(i+1)*c].copy_(h_in[i*c:(i+1)*c], non_blocking=True)
        with torch.cuda.stream(ss[2 + i]): h_out[i*c:(i+1)*c].copy_(d_out[i*c:(i+1)*c], non_blocking=True)
print("both, each direction split over 2 streams  %.1f GB/s each way" % (n / t(both_split) / 1e9))
