import itertools, sys
SROWS=6
def groups_b128():
    return [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
            list(range(32,36))+list(range(44,48))+list(range(52,60)), list(range(36,44))+list(range(48,52))+list(range(60,64))]
def cyc(addrs_per_lane, groups, nbanks, width_dw):
    tot=0
    for g in groups:
        bank={}
        for l in g:
            a=addrs_per_lane[l]
            if a is None: continue
            for d in range(width_dw):
                dw=a//4+d
                bank.setdefault(dw%nbanks,set()).add(dw)
        tot+=max((len(v) for v in bank.values()), default=0) if bank else 0
    return tot
def sim(phys):   # phys(row, slot) -> physical slot index
    total={}
    # dgrad reads: per (dx, u): lane (li, lk): row lk + 2*(u>>1), slot (0)+li+dx+16*(u&1)   [SL0-1 = 0]
    c=0;n=0
    for dx in range(3):
        for u in range(4):
            ad=[phys((l>>4)+2*(u>>1), (l&15)+dx+16*(u&1))*16 for l in range(64)]
            c+=cyc(ad, groups_b128(), 64, 4); n+=1
    total['dgrad_b128']=(c, n*4)
    # commit writes: lane = (l_r, l_seg) for lane<60, pixel e: slot 4*seg+e-3
    c=0;n=0
    for e in range(4):
        ad=[]
        for l in range(64):
            if l>=60: ad.append(None); continue
            r,seg=l//10,l%10; s=4*seg+e-3
            ad.append(phys(r,s)*16 if 0<=s<34 else None)
        c+=cyc(ad,[list(range(8*k,8*k+8)) for k in range(8)],32,4); n+=1
    total['commit_w128']=(c, n*8)
    # wgrad A tr reads: a_off = ((1 + (tq>>1)) + 2rpi) row, slot SL0 + 8lk + tj (+4 for hi), byte 8*(tq&1)
    g2=[list(range(0,32)),list(range(32,64))]
    c=0;n=0
    for rpi in range(2):
        for hi in range(2):
            ad=[]
            for l in range(64):
                li,lk=l&15,l>>4; tj,tq=li>>2,li&3
                ad.append(phys(1+(tq>>1)+2*rpi, 1+8*lk+tj+4*hi)*16+8*(tq&1))
            c+=cyc(ad,g2,64,2); n+=1
    total['wgradA_tr']=(c,n*2)
    c=0;n=0
    for rpi in range(2):
        for nb in range(2):
            for dx in range(3):
                for hi in range(2):
                    ad=[]
                    for l in range(64):
                        li,lk=l&15,l>>4; tj,tq=li>>2,li&3
                        ad.append(phys(tq+2*rpi, 0+8*lk+tj+dx+4*hi)*16+8*nb)
                    c+=cyc(ad,g2,64,2); n+=1
    total['wgradB_tr']=(c,n*2)
    return total
def report(name, phys):
    t=sim(phys)
    # per strip (GC=8): dgrad reads x3 planes; commit x (2 images x 3 planes); A reads x3 planes; B reads x3 planes
    w = t['dgrad_b128'][0]*3 + t['commit_w128'][0]*6 + t['wgradA_tr'][0]*3 + t['wgradB_tr'][0]*3
    b = t['dgrad_b128'][1]*3 + t['commit_w128'][1]*6 + t['wgradA_tr'][1]*3 + t['wgradB_tr'][1]*3
    print(f"{name:40s} total {w:5d} (conflict-free {b}) ", {k:v for k,v in t.items()})
    return w
for RS in (34,35,36,40,48):
    report(f"linear RS={RS}", lambda r,s,RS=RS: r*RS+s)
# per-row rotations within RS=35 (and 34)
import random
best=None
for RS in (34,35):
    for trial in range(4000):
        rot=[0]+[random.randrange(RS) for _ in range(5)]
        f=lambda r,s,RS=RS,rot=rot: r*RS+((s+rot[r])%RS)
        t=sim(f)
        w = t['dgrad_b128'][0]*3 + t['commit_w128'][0]*6 + t['wgradA_tr'][0]*3 + t['wgradB_tr'][0]*3
        if best is None or w<best[0]: best=(w,RS,rot)
print(best)
w,RS,rot=best
report(f"rot RS={RS} {rot}", lambda r,s: r*RS+((s+rot[r])%RS))
print("---- swizzles")
for RS in (34,35):
    for name,fn in [("s^(2*((s>>3)&1))", lambda s: s ^ (2*((s>>3)&1))),
                    ("s^(((s>>3)&3))", lambda s: s ^ ((s>>3)&3)),
                    ("s^(2*((s>>2)&1))", lambda s: s ^ (2*((s>>2)&1))),
                    ("s^(((s>>2)&3))", lambda s: s ^ ((s>>2)&3)),
                    ("s^(((s>>2)&1)|(2*((s>>3)&1)))", lambda s: s ^ (((s>>2)&1)|(2*((s>>3)&1)))),
                    ]:
        # must be a bijection on 0..RS-1 staying < RS (34 -> swizzle may map 32,33 outside: check)
        img=[fn(s) for s in range(34)]
        if len(set(img))!=34 or max(img)>=RS: 
            print(RS,name,"not usable",max(img)); continue
        report(f"RS={RS} {name}", lambda r,s,RS=RS,fn=fn: r*RS+fn(s))
