#!/usr/bin/env python
"""Register copies inside the innermost MFMA loops of a gfx950 assembly listing (`hipcc -S --cuda-device-only`).

    python tools/isa_loop_moves.py kernels.s [--min-mfma 16]

Why: a loop whose accumulators the compiler cannot keep in place (two exits with different live ranges, a phi it fails to
coalesce) gets a copy of every accumulator register per iteration -- register-to-register `v_mov_b64` / `v_accvgpr_*` behind the last MFMA, a dependent
bubble of the matrix pipe.  The 1x1 GEMM of round 6 carried 32 of them per chunk pair; the resource
report does not show it, the listing does.  Prints one line per innermost loop with at least `--min-mfma` MFMAs:
moves, MFMAs, kernel.  tests/test_abi.py runs the same count on the library's two translation units."""
import argparse
import re


def innermost_loop_bodies(text):
    """{(kernel, header label): [instruction lines]} for every innermost loop.  The compiler annotates the header block `This Inner
    Loop Header` and every other block of the loop `in Loop: Header=BBn_m` (on the label line, or on a `; %bb.k:` comment for a
    block without a label); the blocks of a loop need not be contiguous nor in order, so membership is taken from these
    annotations, not from line ranges."""
    loops = {}
    fn, cur = None, None
    for l in text.split('\n'):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            fn, cur = m.group(1), None
            continue
        if fn is None:
            continue
        if 's_endpgm' in l:
            fn, cur = None, None
            continue
        m = re.match(r'^(?:\.L(BB\d+_\d+):|; %bb\.\d+:)(.*)$', l)
        if m:
            note = m.group(2)
            h = re.search(r'in Loop: Header=(BB\d+_\d+)', note)
            if 'This Inner Loop Header' in note and m.group(1):
                cur = (fn, m.group(1))
                loops.setdefault(cur, [])
            elif h and (fn, h.group(1)) in loops:
                cur = (fn, h.group(1))
            elif h:                                            # a block of the loop placed before its header
                cur = (fn, h.group(1))
                loops.setdefault(cur, [])
            else:
                cur = None
            continue
        if cur is not None and l.strip() and not l.strip().startswith(';'):
            loops[cur].append(l)
    return loops


def innermost_mfma_loops(text, min_mfma=16):
    """[(moves, mfmas, kernel)] for every innermost loop with >= min_mfma MFMA instructions."""
    out = []
    for (fn, _), seg in innermost_loop_bodies(text).items():
        mf = sum('v_mfma' in s for s in seg)
        mv = sum(bool(re.search(r'\bv_mov_b64_e32 v\[\d+:\d+\], v\[|\bv_accvgpr_(read|write|mov)', s)) for s in seg)
        if mf >= min_mfma:
            out.append((mv, mf, fn))
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
