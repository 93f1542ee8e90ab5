"""Where does a gfx950 kernel touch scratch?  usage: asm_spills.py file.s <kernel-name-substring>
Reads `hipcc -S --cuda-device-only` output; prints every scratch_load/store with the loops (backward branches) around it,
so that a spill inside the hot loops can be told from per-item state parked outside them.  Works without a GPU."""
import re, sys

def kernel_body(lines, sub):
    start = end = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\S*:", l) and sub in l:
            start = i
        elif start is not None and l.startswith("\t.amdhsa_kernel") or (start is not None and l.startswith(".Lfunc_end")):
            end = i
            break
    return lines[start:end], start

def main():
    path, sub = sys.argv[1], sys.argv[2]
    body, off = kernel_body(open(path).read().split("\n"), sub)
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] <= i:
            loops.append((labels[m.group(2)], i))
    loops.sort()
    def nest(i):
        return [(a, b) for a, b in loops if a <= i <= b]
    n_instr = sum(1 for l in body if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"))
    print("kernel lines", len(body), "instructions", n_instr, "loops", len(loops))
    for a, b in loops:
        cnt = {}
        for l in body[a:b + 1]:
            t = l.strip().split()
            if t and not t[0].startswith((".", ";")):
                k = "valu" if t[0].startswith("v_") else "salu" if t[0].startswith("s_") else t[0].split("_")[0]
                cnt[k] = cnt.get(k, 0) + 1
        print("loop %5d..%5d depth %d  %s" % (a, b, len(nest(a)), cnt))
    for i, l in enumerate(body):
        if "scratch_" in l:
            ns = nest(i)
            print("%5d %-60s loops=%s" % (i, l.strip()[:60], ["%d..%d" % x for x in ns]))

main()
