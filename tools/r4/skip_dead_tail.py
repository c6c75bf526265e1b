"""Ragged batch (configs2_prosody_b8 fixture, the reference's own output): error of the generated frames against the reference with the
padding blocks computed (skip_dead 0) and left out (1), split into each sample's LAST 16 valid frames (what the reference's unmasked
position-embedding conv -- kernel 31, modules.py:167-190 called from dit.py:98 without a mask -- couples to the padding rows) and the rest."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from lemas_tts_amd import synth
import test_gpu_00_sample as T

name = sys.argv[1] if len(sys.argv) > 1 else "configs2_prosody_b8"
gd = os.path.join(R, "tests", "golden")
fx, arch, sd = T._load(gd, name)
fx = synth.expand_reference_fixture(fx)
m = T._model(arch, int(fx["vocab"]), int(fx["wseed"]), bool(fx["prosody"]), sd)
outs = {}
for skip in (0, 1, 2):
    m.engine.set_option("skip_dead", skip)
    out, _ = T._run_case(fx, arch, sd, graph=True, traj=False)
    outs[skip] = np.asarray(out, dtype=np.float64)
    ref = np.asarray(fx["out"], dtype=np.float64)
    tail_se = body_se = 0.0; tail_n = body_n = 0; tail_max = body_max = 0.0
    for b in range(int(fx["B"])):
        L, D = int(fx["lens"][b]), int(fx["duration"][b])
        t0 = max(L, D - 16)
        dt, db = outs[skip][b, t0:D] - ref[b, t0:D], outs[skip][b, L:t0] - ref[b, L:t0]
        tail_se += (dt ** 2).sum(); tail_n += dt.size; body_se += (db ** 2).sum(); body_n += db.size
        tail_max = max(tail_max, np.abs(dt).max()); body_max = max(body_max, np.abs(db).max() if db.size else 0.0)
    print(f"{name} skip_dead={skip}: mel-MSE vs the reference  all generated {T._gen_mse(out, fx['out'], fx):.3e}   last 16 frames of each sample "
          f"{tail_se / tail_n:.3e} (max |err| {tail_max:.3e})   the other generated frames {body_se / max(body_n, 1):.3e} (max |err| {body_max:.3e})")
m.engine.set_option("skip_dead", 0)
m.engine.set_option("skip_masked", 0)
full = np.asarray(T._run_case(fx, arch, sd, graph=True, traj=False)[0], dtype=np.float64)
m.engine.set_option("skip_masked", 1)
print("attention half computed for the padding blocks too (skip_masked 0) vs the default: whole output tensors equal:", bool((full == outs[0]).all()))
d = 0.0; dmax = 0.0; far = 0.0
for b in range(int(fx["B"])):
    L, D = int(fx["lens"][b]), int(fx["duration"][b])
    x = np.abs(outs[0][b, :D] - outs[1][b, :D])
    dmax = max(dmax, x.max()); far = max(far, x[:max(D - 16, 0)].max() if D > 16 else 0.0)
print(f"skip 0 vs skip 1 on the valid frames: max |diff| {dmax:.3e}; on the frames more than 16 before a sample's end {far:.3e}")
