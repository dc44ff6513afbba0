"""Instruction mix of every kernel in a hipcc -S listing (static counts; loops counted once).

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -c X.hip -o X.s && python tools/isa_mix.py X.s [name-substring]
"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    parts = re.split(r"\n(_Z\w+):[^\n]*\n", txt)
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
        if pat not in name:
            continue
        c = collections.Counter()
        for line in body.splitlines():
            line = line.strip()
            if not line or line[0] in ".;/" or line.endswith(":"):
                continue
            c[line.split()[0]] += 1
        g = collections.Counter()
        for op, n in c.items():
            if op.startswith("v_mfma"): g["mfma"] += n
            elif op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq")): g["trans"] += n
            elif op.startswith("v_pk_"): g["v_pk"] += n
            elif op.startswith("v_"): g["valu"] += n
            elif op.startswith("ds_"): g["ds"] += n
            elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): g["vmem"] += n
            elif op.startswith("s_waitcnt"): g["waitcnt"] += n
            elif op.startswith("s_"): g["salu"] += n
            else: g["other"] += n
        print(name[:70], sum(c.values()), dict(g))
        print("   valu:", [(k, v) for k, v in c.most_common(60) if k.startswith("v_") and not k.startswith("v_mfma")][:24])
        print("   ds/vmem:", [(k, v) for k, v in c.items() if k.startswith(("ds_", "buffer_", "global_", "scratch_"))])


if __name__ == "__main__":
    main()
