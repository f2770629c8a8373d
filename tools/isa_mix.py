"""Instruction mix of a kernel's LAST loop nest in a hipcc -S listing (VALU / MFMA / SALU / LDS / VMEM counts).
usage: python tools/isa_mix.py <file.s> <mangled-kernel-name-substring>"""
import collections, re, sys

def main():
    txt = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and ":" in l and "@" in l))
    end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
    body = txt[start:end]
    heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
    loop = body[heads[-1]:] if heads else body
    cnt, valu = collections.Counter(), collections.Counter()
    for l in loop:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m:
            continue
        op = m.group(1)
        k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_")
             else "lds" if op.startswith("ds_") else "vmem" if op.split("_")[0] in ("global", "buffer", "flat", "scratch") else "other")
        cnt[k] += 1
        if k == "valu":
            valu[op] += 1
    print(dict(cnt))
    print(valu.most_common(25))

if __name__ == "__main__":
    main()
