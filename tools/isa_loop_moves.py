#!/usr/bin/env python
"""Register copies inside the innermost MFMA loops of a gfx950 assembly listing (`hipcc -S --cuda-device-only`).

    python tools/isa_loop_moves.py kernels.s [--min-mfma 16]

Why: a loop whose accumulators the compiler cannot keep in place (two exits with different live ranges, a phi it fails to
coalesce) gets a copy of every accumulator register per iteration -- `v_mov_b64` / `v_accvgpr_*` behind the last MFMA, a dependent
bubble of the matrix pipe.  The 1x1 GEMM of round 6 carried 32 of them per chunk pair; the resource
report does not show it, the listing does.  Prints one line per innermost loop with at least `--min-mfma` MFMAs:
moves, MFMAs, kernel.  tests/test_abi.py runs the same count on the library's two translation units."""
import argparse
import re


def innermost_mfma_loops(text, min_mfma=16):
    """[(moves, mfmas, kernel)] for every innermost loop with >= min_mfma MFMA instructions.  A loop = a label the compiler
    annotates `This Inner Loop Header` up to the last branch back to it (a backward branch to an unannotated label is block
    layout, not a loop)."""
    lines = text.split('\n')
    out = []
    fn, headers, last_branch = None, {}, {}

    def flush():
        for lab, a in headers.items():
            if lab not in last_branch:
                continue
            seg = lines[a:last_branch[lab]]
            mf = sum('v_mfma' in s for s in seg)
            mv = sum(bool(re.search(r'\bv_mov_b64|\bv_accvgpr_(read|write|mov)', s)) for s in seg)
            if mf >= min_mfma:
                out.append((mv, mf, fn))

    for ln, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            fn, headers, last_branch = m.group(1), {}, {}
            continue
        if fn is None:
            continue
        m = re.match(r'^(\.LBB\d+_\d+):.*This Inner Loop Header', l)
        if m:
            headers[m.group(1)] = ln
        m = re.search(r'\s(?:s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in headers:
            last_branch[m.group(1)] = ln
        if 's_endpgm' in l:
            flush()
            fn = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('listing')
    ap.add_argument('--min-mfma', type=int, default=16)
    a = ap.parse_args()
    for mv, mf, fn in sorted(innermost_mfma_loops(open(a.listing).read(), a.min_mfma), reverse=True):
        print(mv, mf, fn[:140])


if __name__ == '__main__':
    main()
