"""HBM reference points for the conv tensors (4 x 33.5 MB in, 4 x 33.5 MB out, rotating sets, graph replay)."""
import torch
B, NSETS, REPS = 64, 4, 10
sets = [[(torch.randn(B, 8, 128, 128, device="cuda"), torch.empty(B, 8, 128, 128, device="cuda")) for _ in range(4)] for _ in range(NSETS)]
def bench(fn, tag, nbytes):
    for s in sets:
        fn(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            for _ in range(REPS):
                for s in sets:
                    fn(s)
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (REPS * NSETS)
    print(f"{tag:28s} {us:7.1f} us per set  {nbytes / us / 1e6:5.2f} TB/s")
big_in = [torch.cat([a for a, _ in s]) for s in sets]
big_out = [torch.empty_like(b) for b in big_in]
n = big_in[0].numel() * 4
for i, s in enumerate(sets):
    s.append((big_in[i], big_out[i]))
bench(lambda s: s[4][1].copy_(s[4][0]), "copy 134 MB -> 134 MB (1 launch)", 2 * n)
bench(lambda s: [o.copy_(a) for a, o in s[:4]], "copy 4 x (33.5 -> 33.5) MB", 2 * n)
bench(lambda s: s[4][1].fill_(1.0), "fill 134 MB", n)
bench(lambda s: torch.relu_(s[4][0]), "in-place relu 134 MB (r+w)", 2 * n)
