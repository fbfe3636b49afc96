"""Why do two query batches in flight (two host threads) gain so little in one process?  Phase times of
every step (embed / retrieve+gather / register, host clock, each phase ended by a stream sync) run
sequentially and with two threads; plus the same with two PROCESSES when argv[1] == 'proc'."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from corsair_amd import _lib, harness, synth, registration
dev = torch.device("cuda:0")
cfg = harness.Config()
sd, emb = synth.make_state_dicts(cfg.random_seed)
pipe = harness.Pipeline(sd, emb, device=dev, config=cfg)
C, BATCH, STEPS = 652, 32, 8
cat = pipe.embed_clouds([synth.make_cloud(c, 15000)[:cfg.n_points] for c in range(C)])
sym = np.ones(C, np.int32); sym[::326] = 4
q_dev, q_off = [], []
for b in range(STEPS):
    chunk = [synth.apply_pose(synth.make_cloud((b * BATCH + i) % C, 15000)[15000 - cfg.n_points:], synth.random_pose(b * BATCH + i, max_trans=0.0)) for i in range(BATCH)]
    q_dev.append(torch.from_numpy(np.concatenate(chunk, 0)).to(dev))
    q_off.append(np.concatenate([[0], np.cumsum([len(c) for c in chunk])]).tolist())
log = []
def step(b, tag):
    st = torch.cuda.current_stream()
    t0 = time.perf_counter()
    qs = pipe.embed_batch(q_dev[b], q_off[b]); st.synchronize()
    t1 = time.perf_counter()
    ids = [(2 * (b * BATCH + i), 2 * (b * BATCH + i) + 1) for i in range(BATCH)]
    q_anc = [registration.draw_anchors(qs.offsets[i + 1] - qs.offsets[i], 100, ids[i][0]) for i in range(BATCH)]
    top = _lib.to_host(pipe.retrieve(qs.desc, cat.desc, 1)[:, 0])[0]
    cads = cat.gather(top); st.synchronize()
    t2 = time.perf_counter()
    res = pipe.register(qs, cads, sym[top], anchor_ids=ids, force_gate=True, query_anchors=q_anc)
    _lib.to_host(res.T_best, res.T_ransac, res.cd_best, res.iters)
    t3 = time.perf_counter()
    log.append((tag, b, t0, t1, t2, t3))
def run(depth):
    log.clear()
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    def worker(w):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[w]):
            for b in range(w, STEPS, depth):
                step(b, w)
    torch.cuda.synchronize(); t = time.perf_counter()
    th = [threading.Thread(target=worker, args=(w,)) for w in range(depth)]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); el = time.perf_counter() - t
    e = np.mean([l[3] - l[2] for l in log]) * 1e3; r = np.mean([l[4] - l[3] for l in log]) * 1e3; g = np.mean([l[5] - l[4] for l in log]) * 1e3
    print("depth %d: %.1f ms per step wall; per step embed %.2f ms, retrieve+gather %.2f ms, register %.2f ms (sum %.2f)" % (depth, el / STEPS * 1e3, e, r, g, e + r + g), flush=True)
run(1); run(1); run(2); run(2); run(1)
