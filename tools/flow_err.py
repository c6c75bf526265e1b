"""Development aid: relative rms error of one CFG flow evaluation (HIP vs fp32 oracle) for a few (depth, N, t)."""
import sys
sys.path.insert(0, ".")
import torch
from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import DiTArch
from lemas_tts_amd.model.cfm import CFM
from oracle import lemas_oracle as O

torch.set_num_threads(16)
VOCAB = 898
for depth, N, F_ in ((2, 1875, 938), (22, 400, 150), (22, 1875, 938)):
    arch = DiTArch(depth=depth)
    sd = synth.synth_cfm_state_dict(arch, VOCAB, 1234)
    cond = torch.from_numpy(synth.synth_cond_mel(1, F_))[None]
    text = torch.from_numpy(synth.synth_tokens(2, round(N * 0.17), VOCAB))[None]
    y0 = torch.from_numpy(synth.synth_noise(3, N))[None]
    m = CFM(arch, VOCAB, sd, device="cuda:0")
    cmask = torch.zeros(1, N, dtype=torch.bool); cmask[:, :F_] = True
    cpad = torch.nn.functional.pad(cond, (0, 0, 0, N - F_))
    grid = O.time_grid(32, 5)
    m.engine.prepare(cpad, cmask, text, grid.numpy(), cond_frames=F_, cfg_strength=2.0)
    oc = O.OracleCFM(sd, arch)
    sc = torch.where(cmask[..., None], cpad, torch.zeros_like(cpad))
    for k in (0, 16, 31):
        pred = m.engine.forward(y0, k).cpu()
        rc = oc.dit.forward(y0, sc, text, grid[k], False, False, None, True)
        ru = oc.dit.forward(y0, sc, text, grid[k], True, True, None, True)
        e_c = float((pred[0] - rc[0]).pow(2).mean().sqrt() / rc[0].pow(2).mean().sqrt())
        e_u = float((pred[1] - ru[0]).pow(2).mean().sqrt() / ru[0].pow(2).mean().sqrt())
        cg = 2.0 * (1 - float(grid[k])) ** 2
        fh = pred[0] + (pred[0] - pred[1]) * cg
        fr = rc[0] + (rc[0] - ru[0]) * cg
        e_f = float((fh - fr).pow(2).mean().sqrt() / fr.pow(2).mean().sqrt())
        print(f"depth {depth:2d} N {N:4d} step {k:2d} t={float(grid[k]):.3f}: rel err cond {e_c:.3e} uncond {e_u:.3e} cfg-flow {e_f:.3e} "
              f"|pred| {float(rc.pow(2).mean().sqrt()):.3f} |pred-null| {float((rc - ru).pow(2).mean().sqrt()):.3f}")
    oc.dit.clear_cache()
    del m
