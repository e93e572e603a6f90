"""Static scan of a gfx950 listing (hipcc -S --cuda-device-only) for vector-memory loads whose result the wave waits for almost
immediately -- the pattern behind round 5's head-kernel finding: a prefetch whose lambda converts / compares the loaded value makes the
compiler place `s_waitcnt vmcnt(..)` right behind the load, and the full memory latency is exposed once per loop iteration.

For every kernel: walks the instruction stream in program order, keeps the FIFO of outstanding vector-memory operations (loads and
stores both count in vmcnt on gfx9), and at every `s_waitcnt vmcnt(N)` reports the youngest LOAD it forces to complete together with the
number of instructions issued since that load.  Branches are ignored (straight-line approximation): read the report as a list of places to
look at, not as a proof.     usage: python tools/isa_exposed_loads.py /tmp/head.s [--max-dist 24] [--loops-only]"""
import re
import sys


def kernels(lines):
    cur, name = None, None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, name = i, m.group(1)
        if ".amdhsa_kernel" in l and cur is not None:
            yield name, lines[cur:i]
            cur = None


def demangle_short(n):
    m = re.search(r"\d+([a-z_0-9]+kernel)", n)
    tail = re.findall(r"ILi(\d+)E", n)
    b = re.findall(r"ILb(\d)E", n)
    return (m.group(1) if m else n[:60]) + ("<" + ",".join(re.findall(r"L[ib](\d+)E", n)) + ">" if (tail or b) else "")


def scan(body, max_dist, loops_only):
    out = []
    fifo = []          # (instruction index, is_load, text)
    n = 0
    in_loop = False
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if "Loop Header" in l or "in Loop" in l:
                in_loop = True
            elif re.match(r"^\.LBB\d+_\d+:\s*$", l):
                in_loop = False
            continue
        op = t.split()[0]
        n += 1
        if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            fifo.append((n, True, t))
        elif op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
            fifo.append((n, False, t))
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                keep = int(m.group(1))
                done, fifo = (fifo[:len(fifo) - keep], fifo[len(fifo) - keep:]) if keep < len(fifo) else ([], fifo)
                loads = [d for d in done if d[1]]
                if loads:
                    idx, _, text = loads[-1]
                    if n - idx <= max_dist and (in_loop or not loops_only):
                        out.append((n - idx, in_loop, text[:70], t))
    return out


def main():
    path = sys.argv[1]
    max_dist = int(sys.argv[sys.argv.index("--max-dist") + 1]) if "--max-dist" in sys.argv else 24
    loops_only = "--loops-only" in sys.argv
    lines = open(path).read().splitlines()
    for name, body in kernels(lines):
        hits = scan(body, max_dist, loops_only)
        if hits:
            print(f"{demangle_short(name)}  ({len(body)} lines)")
            for d, lp, text, w in hits:
                print(f"    {'loop' if lp else '    '}  {d:3d} instr after  {text:70s}  {w}")


if __name__ == "__main__":
    main()
