"""Where fp32 rollouts leave the fp64 ones: the cases of tools/parity_sweep.py replayed on the host build of the
kernel arithmetic (tests/host_core_harness.cpp, same source as the device code) in both precisions, substep by
substep.  For every sample: the first substep at which the states differ by more than a threshold, the rows whose
activation differs there, and how the difference grew.  CPU only.

    python tools/outlier_trace.py [--as-swept] [case ...]        (default: cases 4 and 22)

--as-swept draws the cases the way the round-2 sweep did (profiles/r2_parity_sweep.txt flags its cases 4 and 22)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sweep_cases import cases                   # noqa: E402
from trajopt_amd.model import get_model         # noqa: E402

SRC, OUT = os.path.join(ROOT, "tests", "host_core_harness.cpp"), os.path.join(ROOT, "tests", "_host_core.so")
deps = [SRC] + [os.path.join(ROOT, "trajopt_amd", "csrc", f) for f in ("dyn_core.hpp", "env_core.hpp", "models_gen.hpp")]
if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", OUT, SRC])
L = C.CDLL(OUT)
md = get_model("reacher_7dof")
lo, hi = np.array(md.range_lo), np.array(md.range_hi)
K, H = 256, 32
AS_SWEPT_IN_ROUND_2 = "--as-swept" in sys.argv


def sweep_cases(last):
    """The sweep's cases 0 .. last (tests/sweep_cases.py): (case, q0, v0, actions)."""
    for case, q0, v0, tgt, raw, base, act in cases(md, last + 1, K, H, AS_SWEPT_IN_ROUND_2):
        yield case, q0, v0, act


def trace(f64, q0, v0, act):
    out = np.zeros((2 * H, 24))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.core_trace(int(f64), p(np.ascontiguousarray(q0)), p(np.ascontiguousarray(v0)), p(np.ascontiguousarray(act)), H, p(out))
    return out


WIDTH = 0.001            # solimp width of the model: the impedance rises over the first milliradian of a violation


def jump_cause(a, b, err):
    """What stands at the largest multiplicative jump of the difference between the precisions: (substep, cause)."""
    ratio = err[1:] / np.maximum(err[:-1], 1e-9)
    s = int(np.argmax(ratio)) + 1                       # the substep whose result jumped
    pen_a = np.minimum(a[:, :7] - lo, hi - a[:, :7])
    pen_b = np.minimum(b[:, :7] - lo, hi - b[:, :7])
    w = slice(max(0, s - 2), s + 1)                     # states the substep started from and ended in
    if ((pen_a[w] < 0) != (pen_b[w] < 0)).any():
        return s, "row exists in one precision only"
    if ((a[w, 14:22] > 0) != (b[w, 14:22] > 0)).any():
        return s, "row active in one precision only"
    if ((pen_a[w] < 0) & (pen_a[w] > -WIDTH)).any() or ((pen_b[w] < 0) & (pen_b[w] > -WIDTH)).any():
        return s, "row inside the impedance ramp"
    return s, "smooth"


def survey():
    """All 40 cases: per case the number of samples that end more than 1e-4 apart and what stands at the largest jump
    of the difference.  --exact-positions: the fp32 side integrates its positions without rounding (see the harness)."""
    f32 = 2 if "--exact-positions" in sys.argv else 0
    causes = ["row exists in one precision only", "row active in one precision only", "row inside the impedance ramp", "smooth"]
    tot = dict({"samples": 0, "apart": 0}, **{c: 0 for c in causes})
    finals = []
    for case, q0, v0, act in sweep_cases(39):
        apart = []
        for k in range(K):
            a, b = trace(1, q0, v0, act[k]), trace(f32, q0, v0, act[k])
            err = np.abs(a[:, :14] - b[:, :14]).max(axis=1) / (1 + np.abs(a[:, :14]).max(axis=1))
            finals.append(err[-1])
            if err[-1] <= 1e-4:
                continue
            apart.append(jump_cause(a, b, err)[1])
        tot["samples"] += K
        tot["apart"] += len(apart)
        for c in apart:
            tot[c] += 1
        pen = np.maximum(lo - q0, q0 - hi).max()
        if apart:
            print("case %2d  deepest start violation %.3f rad, |v0| max %.1f: %d of %d samples end > 1e-4 apart: %s"
                  % (case, max(pen, 0.0), np.abs(v0).max(), len(apart), K, {c: apart.count(c) for c in set(apart)}), flush=True)
    print("all cases:", tot, " final relative state difference: median %.2e, 99th percentile %.2e, max %.2e"
          % (np.median(finals), np.percentile(finals, 99), np.max(finals)))


def main():
    if "--survey" in sys.argv:
        return survey()
    want = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [4, 22]
    for case, q0, v0, act in sweep_cases(max(want)):
        if case not in want:
            continue
        print("case %d: q0 - lo = %s\n         hi - q0 = %s" % (case, np.array2string(q0 - lo, precision=4), np.array2string(hi - q0, precision=4)))
        first, cause, final = [], [], []
        for k in range(K):
            a, b = trace(1, q0, v0, act[k]), trace(0, q0, v0, act[k])
            err = np.abs(a[:, :14] - b[:, :14]).max(axis=1) / (1 + np.abs(a[:, :14]).max(axis=1))
            rows_a, rows_b = a[:, 14:22] > 0, b[:, 14:22] > 0
            # row existence: the joint is outside its range after the substep
            out_a = (a[:, :7] < lo) | (a[:, :7] > hi)
            out_b = (b[:, :7] < lo) | (b[:, :7] > hi)
            final.append(err[-1])
            big = np.nonzero(err > 1e-5)[0]
            if len(big) == 0:
                first.append(-1); cause.append("none")
                continue
            s = big[0]
            first.append(s)
            w = slice(max(0, s - 3), s + 1)
            if (out_a[w] != out_b[w]).any():
                j = np.nonzero((out_a[w] != out_b[w]).any(axis=0))[0]
                cause.append("a limit row exists in one precision only (joint %s)" % j)
            elif (rows_a[w] != rows_b[w]).any():
                j = np.nonzero((rows_a[w] != rows_b[w]).any(axis=0))[0]
                cause.append("a row is active in one precision only (row %s)" % j)
            else:
                cause.append("smooth growth")
        first, final = np.array(first), np.array(final)
        print("  samples with a relative state difference > 1e-5 at some substep: %d of %d; > 1e-4 at the end: %d; median final %.2e, max %.2e"
              % ((first >= 0).sum(), K, (final > 1e-4).sum(), np.median(final), final.max()))
        for c in sorted(set(cause)):
            idx = [i for i, x in enumerate(cause) if x == c]
            print("  %-70s %4d samples, final difference median %.2e max %.2e" % (c, len(idx), np.median(final[idx]), final[idx].max()))
        # the growth of the difference for the worst sample
        kw = int(np.argmax(final))
        a, b = trace(1, q0, v0, act[kw]), trace(0, q0, v0, act[kw])
        err = np.abs(a[:, :14] - b[:, :14]).max(axis=1)
        print("  worst sample %d: |state difference| by substep:" % kw, " ".join("%.0e" % e for e in err[::4]))
        print("     joints outside their range (fp64), substep 0:", np.nonzero((a[0, :7] < lo) | (a[0, :7] > hi))[0],
              " penetration at substep 0:", np.array2string(np.minimum(a[0, :7] - lo, hi - a[0, :7]), precision=5))


if __name__ == "__main__":
    main()
