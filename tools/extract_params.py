#!/usr/bin/env python3
"""Extract the *parameters* (not code) of the hot path from the read-only reference checkout.

Runs only in the build container (where /root/reference exists); its outputs are small data files that are
committed:  oracle/params/poseidon1_rc.inc  and  leanmultisig_amd/csrc/params/poseidon1_rc.inc
(identical content: the 28x16 canonical round constants of Poseidon1-16 over KoalaBear,
 reference: crates/backend/koala-bear/src/poseidon1_koalabear_16.rs:699-815).
"""
import re, sys, os

REF = "/root/reference/crates/backend/koala-bear/src/poseidon1_koalabear_16.rs"

def main():
    src = open(REF).read()
    start = src.index("const POSEIDON1_RC")
    end = src.index("]);", start)
    body = src[start:end]
    body = body[body.index("new_2d_array("):]
    vals = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", body)]
    assert len(vals) == 28 * 16, len(vals)
    lines = ["/* Poseidon1-16 KoalaBear round constants, canonical (non-Montgomery) values.",
             " * DATA extracted by tools/extract_params.py from the reference",
             " * crates/backend/koala-bear/src/poseidon1_koalabear_16.rs:699-815 (28 rounds x 16 lanes). */"]
    for r in range(28):
        row = ", ".join("0x%08xu" % v for v in vals[16 * r:16 * r + 16])
        lines.append("{ %s }," % row)
    out = "\n".join(lines) + "\n"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d in ("oracle/params", "leanmultisig_amd/csrc/params"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
        open(os.path.join(root, d, "poseidon1_rc.inc"), "w").write(out)
    print("wrote", len(vals), "constants")

if __name__ == "__main__":
    main()
