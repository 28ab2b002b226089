#!/usr/bin/env python
"""Condense the output of tools/ubench/bin/mfma_pk_hazard: every row with misses, and per section the list of clean rows
reduced to their parameters (so that "clean" is a statement about named configurations, not an omission).

    tools/ubench/bin/mfma_pk_hazard | python tools/hazard_report.py > profiles/rNN_mfma_pk_hazard.txt
"""
import re
import sys

CLEAN = (re.compile(r"\| 0 0 \| 0 0 \| 0 0 \| 0 0\s*$"), re.compile(r"misses\s+0\s*$"))


def main():
    rows = [l.rstrip("\n") for l in sys.stdin if l.strip()]
    print("# tools/ubench/mfma_pk_hazard.hip on one MI355X (gfx950, ROCm 7.2).  Sections: (1) same wave, one wave per SIMD: MFMA -> s_nop x N ->")
    print("# a dependent pair of v_pk_fma_f32 op_sel:[0,1,0]; pk -> MFMA -> s_nop x N -> pk; pk -> s_nop x N -> MFMA -> pk; (2) cross-wave: chain waves next to partner waves on the same SIMDs (2000 x 64 chain")
    print("# steps per lane; misses counted per lane group and result half); (3) WAR: VALU overwrites an MFMA source N wait states after")
    print("# issue; (4) RAW: VALU reads an MFMA result N wait states after issue (the compiler inserts s_nop 7 = 8).")
    print("# Counts are 32-bit sums over all lanes of all workgroups and wrap for the largest grids of section 4.")
    clean = []
    for l in rows:
        if l.startswith("=="):
            print(l)
            continue
        if any(c.search(l) for c in CLEAN):
            clean.append(re.sub(r":? misses.*$", "", l).strip())
            continue
        print(l)
    print(f"\n# {len(clean)} configurations without a single miss:")
    for c in clean:
        print("#   ", c)


if __name__ == "__main__":
    main()
