"""Report of the bf16-piece convolution experiment (CS_CONV_SPLIT, off by default; VERDICT r3 #10, SURVEY 8d "bf16 only
behind a parity-checked flag"):  python tools/conv_split_report.py [chair|stress]

For the exact f32 chain and for CS_CONV_SPLIT=3 / 2 on the same batch and the same (random-init) network:
  * sum of the convolution launches of one forward (event time per launch, launches serialised) and the useful TF/s,
  * wall time of whole forwards with prebuilt maps,
  * max |difference| of the 16-d per-voxel features and of the 256-d (normalised) descriptors against the exact path,
  * how many top-10 catalog IDs and how many 5-NN correspondence rows change.
The exact path stays the default, the headline and the only path the parity tests cover."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WL = sys.argv[1] if len(sys.argv) > 1 else "chair"
NB, NP, VOX = (64, 15000, 0.02) if WL == "stress" else (32, 10000, 0.03)
os.environ["CS_CONV_SPLIT_CACHE"] = "1"
import numpy as np, torch
from corsair_amd import backend as B, engine, synth, harness

dev = torch.device("cuda:0")
sd, emb = synth.make_state_dicts(31)
pipe = harness.Pipeline(sd, emb, device=dev)
clouds = [synth.make_cloud(c, 15000)[:NP] for c in range(NB)]
cat_clouds = [synth.make_cloud(1000 + c, 15000)[:NP] for c in range(64)]


def prep(cl):
    xyz = torch.from_numpy(np.concatenate(cl)).to(dev)
    off = np.concatenate([[0], np.cumsum([len(c) for c in cl])]).tolist()
    keep, grid, out_off = B.voxelize(xyz, off, VOX)
    feats = torch.ones((grid.shape[0], 1), device=dev)
    return grid, feats, engine.BatchMaps(grid), out_off


grid, feats, maps, voff = prep(clouds)
cgrid, cfeats, cmaps, cvoff = prep(cat_clouds)
print(f"# workload {WL}: {NB} clouds x {NP} points at {VOX} m -> rows", maps.c1.n, maps.c2.n, maps.c4.n, maps.c8.n,
      "pairs", maps.total_pairs())
orig = B.conv_fwd


def run(mode):
    if mode:
        os.environ["CS_CONV_SPLIT"] = str(mode)
    else:
        os.environ.pop("CS_CONV_SPLIT", None)
    log = []

    def timed(kmap, x, weight, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(kmap, x, weight, *a, **k); e1.record(); torch.cuda.synchronize()
        cin, cout = weight.shape[-2], weight.shape[-1]
        pairs = kmap.num_pairs if kmap is not None else x.shape[0]
        log.append((cin, cout, pairs, e0.elapsed_time(e1)))
        return r

    pipe.engine.forward(grid, feats, maps); torch.cuda.synchronize()      # warm (cuts the weights in cache mode)
    B.conv_fwd = timed; engine.B.conv_fwd = timed
    out, f8, _ = pipe.engine.forward(grid, feats, maps)
    g = pipe.engine.embed(f8, maps, NB)
    B.conv_fwd = orig; engine.B.conv_fwd = orig
    conv_ms = sum(l[3] for l in log)
    flop = sum(2.0 * l[2] * l[0] * l[1] for l in log)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        pipe.engine.forward(grid, feats, maps)
    e1.record(); torch.cuda.synchronize()
    cout_, cf8, _ = pipe.engine.forward(cgrid, cfeats, cmaps)
    cg = pipe.engine.embed(cf8, cmaps, len(cat_clouds))
    top = B.l2_topk(g, cg, 10).cpu().numpy()
    # 5-NN of the first cloud's voxel features in the first catalog cloud's
    a0, a1 = int(voff[0]), int(voff[1])
    c0, c1 = int(cvoff[0]), int(cvoff[1])
    nn = B.knn_feat(out[a0:a1].contiguous(), [0, a1 - a0], cout_[c0:c1].contiguous(), [0, c1 - c0], 5).cpu().numpy()
    os.environ.pop("CS_CONV_SPLIT", None)
    return dict(conv_ms=conv_ms, tf=flop / conv_ms / 1e9, fwd_ms=e0.elapsed_time(e1) / 5, out=out.cpu().numpy(),
                g=g.cpu().numpy(), top=top, nn=nn, layers=log)


ref = run(0)
print(f"exact f32 chain : conv {ref['conv_ms']:8.3f} ms per forward = {ref['tf']:6.1f} useful TF/s (f32 matrix peak 157.3); "
      f"forward {ref['fwd_ms']:.3f} ms")
for mode in (3, 2):
    r = run(mode)
    dout = float(np.abs(r["out"] - ref["out"]).max())
    dg = float(np.abs(r["g"] - ref["g"]).max())
    top_flip = int((r["top"] != ref["top"]).sum())
    nn_rows = int((r["nn"] != ref["nn"]).any(axis=1).sum())
    print(f"CS_CONV_SPLIT={mode} : conv {r['conv_ms']:8.3f} ms per forward = {r['tf']:6.1f} useful TF/s ({ref['conv_ms'] / r['conv_ms']:.2f}x); "
          f"forward {r['fwd_ms']:.3f} ms ({ref['fwd_ms'] / r['fwd_ms']:.2f}x); max |d| voxel features {dout:.3e}, descriptors {dg:.3e}; "
          f"top-10 IDs changed {top_flip} of {ref['top'].size}; 5-NN rows changed {nn_rows} of {ref['nn'].shape[0]}")
    if mode == 3:
        print("#   per layer (cin -> cout, pairs, exact us, split us):")
        for (ci, co, pr, ms0), (_, _, _, ms1) in zip(ref["layers"], r["layers"]):
            print(f"#   {ci:4d}->{co:4d} pairs {pr:9d} {ms0 * 1e3:8.1f} {ms1 * 1e3:8.1f}  {ms0 / ms1:5.2f}x")
