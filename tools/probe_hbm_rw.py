"""HBM read / write / copy bandwidth of plain streaming kernels on this box (torch fill / sum / copy over 4 GiB): the
yardstick for the write-heavy backward-data kernels (LFF / GFF.0 dgrad write 2-12x what they read)."""
import time
import torch
dev = torch.device("cuda")
n = 1 << 30                                   # 4 GiB of fp32
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
x.normal_()


def rate(fn, nbytes, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e12


print(f"write only (fill_)      : {rate(lambda: y.fill_(1.5), 4 * n):.2f} TB/s")
print(f"read only (sum)         : {rate(lambda: x.sum(), 4 * n):.2f} TB/s")
print(f"copy (read + write)     : {rate(lambda: y.copy_(x), 8 * n):.2f} TB/s of traffic")
print(f"1 read : 3 writes (3 fills + 1 sum interleaved not possible in one kernel; see copy / fill above)")
h = x[: n // 2].half()
z = torch.empty(n // 2, 4, dtype=torch.float16, device=dev)
print(f"expand 1 -> 4 (fp16)    : {rate(lambda: z.copy_(h[:, None].expand(-1, 4)), (n // 2) * 2 * 5):.2f} TB/s of traffic (1 read : 4 writes)")
