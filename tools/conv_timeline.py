"""Per-workgroup timeline of one grouped conv3x3 launch (start / end wall clock, XCC / CU ids).  GPU only."""
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from popcorn_amd import ops, _lib as L
lib = L.lib()
B = 64
cin, cout, hw = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 8, 128)
sets = []
for s in range(4):
    probs = []
    for i in range(4):
        ca = cin if cin <= 8 else cin // 2
        a = torch.randn(B, ca, hw, hw, device="cuda")
        b = torch.randn(B, cin - ca, hw, hw, device="cuda") if cin > ca else None
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
        bias = torch.zeros(cout, device="cuda")
        probs.append({"a": a, "b": b, "w": w, "bn": L.bn(bias), "out": torch.empty(B, cout, hw, hw, device="cuda"), "_k": bias})
    sets.append(probs)
ts = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
for s in sets:
    ops.conv3x3_fwd_group(s)
torch.cuda.synchronize()
lib.pc_debug_conv(int(os.environ.get('ABL_DBG', '0'), 0), 0)
lib.pc_debug_conv_ts(C.c_void_p(ts.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ops.conv3x3_fwd_group(sets[0]); ops.conv3x3_fwd_group(sets[1])
e0.record()
ops.conv3x3_fwd_group(sets[2])
e1.record()
torch.cuda.synchronize()
lib.pc_debug_conv_ts(C.c_void_p(0))
t = ts.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0          # 100 MHz wall clock -> us
hw_id = t[:, 2] & 0xffffffff
xcc = (t[:, 2] >> 32) & 0xf
cu = (hw_id >> 8) & 0xf; sh = (hw_id >> 12) & 1; se = (hw_id >> 13) & 7
print(f"{cin}->{cout}@{hw}: {len(t)} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us")
print("start us: min %.1f  p50 %.1f  p90 %.1f  max %.1f" % (st.min(), np.percentile(st, 50), np.percentile(st, 90), st.max()))
print("end   us: min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % (en.min(), np.percentile(en, 10), np.percentile(en, 50), np.percentile(en, 90), en.max()))
print("life  us: min %.1f  p50 %.1f  max %.1f   mean %.1f" % ((en - st).min(), np.percentile(en - st, 50), (en - st).max(), (en - st).mean()))
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  xcc {x}: n={m.sum():4d} start p50 {np.percentile(st[m], 50):6.1f} max {st[m].max():6.1f}  end p50 {np.percentile(en[m], 50):6.1f} max {en[m].max():6.1f}  distinct (se,sh,cu) {len(set(zip(se[m], sh[m], cu[m])))}")
key = list(zip(xcc, se, sh, cu))
import collections
cnt = collections.Counter(key)
print("workgroups per CU: ", collections.Counter(cnt.values()))
late = st > 5
print("late starters (>5 us):", late.sum(), " their lifetime mean %.1f" % ((en - st)[late].mean() if late.any() else 0))
prob = np.arange(len(t)) // (len(t) // 4)
for q in range(4):
    print(f"problem {q}: end mean {en[prob == q].mean():.1f}  min {en[prob == q].min():.1f} max {en[prob == q].max():.1f}")
bycu = collections.defaultdict(list)
for i, k in enumerate(key):
    bycu[k].append((round(float(en[i]), 1), int(prob[i])))
for k in list(bycu)[:8]:
    print(k, sorted(bycu[k]))
last = np.array([max(e for e, _ in v) for v in bycu.values()]); first = np.array([min(e for e, _ in v) for v in bycu.values()])
print("per CU: first finisher mean %.1f, last finisher mean %.1f (min %.1f max %.1f)" % (first.mean(), last.mean(), last.min(), last.max()))
