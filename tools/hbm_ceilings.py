"""Dev aid (GPU box): what this MI355X moves with plain torch kernels -- write-only (fill), read-only (sum), copy -- on buffers of
config 5's output size (17.2 GB) and of config 4's (1.6 GB): the practical ceilings the HBM-bound kernels are held against."""
import json, torch
dev = torch.device("cuda:0")
res = {}
for name, n in (("17.2GB", 256 * 8192 * 2048), ("1.6GB", 256 * 2048 * 768)):
    a = torch.empty(n, dtype=torch.float32, device=dev); b = torch.empty_like(a)
    a.normal_()
    def t(f, reps=10):
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    fill = t(lambda: b.fill_(1.5)); rd = t(lambda: a.sum()); cp = t(lambda: b.copy_(a))
    gb = n * 4 / 1e9
    res[name] = {"fill_ms": fill, "fill_TBps": gb / fill, "sum_ms": rd, "sum_TBps": gb / rd, "copy_ms": cp, "copy_TBps_read_plus_write": 2 * gb / cp}
    del a, b
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
