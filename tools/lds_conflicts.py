"""Bank-conflict model of the in-LDS FFT stages (MI355X_MICROARCH.md §LDS): for every LDS instruction of a stage, the lanes of a
wave are served in groups; within a group an access to a busy bank costs an extra cycle.  Compares padding functions.
   python tools/lds_conflicts.py"""
import itertools

def stages(lgn, maxlg=4):
    out, rem = [], lgn
    def lv(r): n = (r + maxlg - 1) // maxlg; return (r + n - 1) // n
    while rem > 0:
        lg = lv(rem); out.append((lg, rem - 1, rem - lg)); rem -= lg      # (LG, LGH top span, lghmin)
    return out

def cost(addrs_elems, elem_dwords, group, nbanks):
    """addrs of one wave-instruction in element units -> (ideal cycles, actual cycles)"""
    ideal = actual = 0
    for g in range(0, 64, group):
        lanes = addrs_elems[g:g + group]
        per_bank = {}
        for a in set(lanes):
            for d in range(elem_dwords):
                per_bank.setdefault((a * elem_dwords + d) % nbanks, set()).add(a)
        actual += max(len(v) for v in per_bank.values())
        ideal += 1
    return ideal, actual

def analyse(lgn, pad, elem_dwords=2, S=4, maxlg=4, ld_extra=1):
    N = 1 << lgn
    LD = pad(N) + ld_extra
    tot = {"read": [0, 0], "write": [0, 0]}
    for (LG, LGH, lghmin) in stages(lgn, maxlg):
        r, lgnb = 1 << LG, lgn - LG
        nq = S << lgnb
        for w0 in range(0, min(nq, 512), 64):
            qs = range(w0, min(w0 + 64, nq))
            if len(qs) < 64:
                continue
            for m in range(r):
                addrs = []
                for q in qs:
                    seq, rr = q >> lgnb, q & ((1 << lgnb) - 1)
                    blk, j = rr >> lghmin, rr & ((1 << lghmin) - 1)
                    addrs.append(seq * LD + pad((blk << (LGH + 1)) + j + (m << lghmin)))
                i, a = cost(addrs, elem_dwords, 32, 64); tot["read"][0] += i; tot["read"][1] += a
                i, a = cost(addrs, elem_dwords, 16, 32); tot["write"][0] += i; tot["write"][1] += a
    return tot

pads = {
    "i + i>>4 (current)": lambda i: i + (i >> 4),
    "i + i>>5": lambda i: i + (i >> 5),
    "i + i>>3": lambda i: i + (i >> 3),
    "i + i>>4 + i>>8": lambda i: i + (i >> 4) + (i >> 8),
    "i + i>>5 + i>>9": lambda i: i + (i >> 5) + (i >> 9),
    "i + 2*(i>>5)": lambda i: i + 2 * (i >> 5),
    "i + i>>4 + i>>7": lambda i: i + (i >> 4) + (i >> 7),
    "i + i>>3 + i>>6 + i>>9": lambda i: i + (i >> 3) + (i >> 6) + (i >> 9),
    "none": lambda i: i,
}
for lgn, S, maxlg in ((10, 4, 4), (9, 4, 4), (10, 2, 4), (10, 4, 3)):
    print(f"N = 2^{lgn}, {S} sequences, cap {maxlg} levels/stage, fp32 complex (2 dwords)")
    for name, f in pads.items():
        for ex in (1,):
            t = analyse(lgn, f, 2, S, maxlg, ex)
            print(f"   {name:26s} read x{t['read'][1] / t['read'][0]:5.2f}   write x{t['write'][1] / t['write'][0]:5.2f}")

if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "search":
        best = []
        for a in range(2, 7):
            for x in range(1, 4):
                for b in range(a + 1, 10):
                    for y in range(0, 4):
                        if y == 0 and b != a + 1:
                            continue
                        f = lambda i, a=a, x=x, b=b, y=y: i + x * (i >> a) + y * (i >> b)
                        c = 0
                        for lgn in (10, 9):
                            t = analyse(lgn, f, 2, 4, 4, 1)
                            c += 2 * t["read"][1] / t["read"][0] + 6 * t["write"][1] / t["write"][0]
                        foot = f(1024) / 1024
                        if y == 0 and b != a + 1:
                            continue
                        best.append((c, foot, a, x, b, y))
        best.sort()
        for c, foot, a, x, b, y in best[:12]:
            print(f"cost {c:6.2f}  footprint x{foot:.3f}   i + {x}*(i>>{a}) + {y}*(i>>{b})")
