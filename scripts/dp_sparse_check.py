"""2-rank self-check (run with GOAT_DIST_BACKEND-independent gloo on one GPU):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29548 scripts/dp_sparse_check.py
The sparse exchange of the word-embedding gradient and the two-phase backward must give the same averaged gradients as the
plain path (one backward, dense all-reduce of everything)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.distributed as dist
import vln_goat_amd
from vln_goat_amd import config as gcfg, dp, pretrain_model, synth

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo')
cfg = gcfg.make_config(num_l_layers=2, num_top_layer=2, num_pano_layers=1, vocab_size=1000)
torch.manual_seed(0)
model = pretrain_model.GlocalTextPathCMTPreTraining(cfg).cuda().eval()
vln_goat_amd.set_compute_dtype(torch.bfloat16)
# instruction lengths differ between the ranks (the collate pads to the per-batch maximum: 30 vs 27 tokens, i.e. 120 vs 108 token
# rows in the sparse exchange of the word-embedding gradient)
gb = synth.batch_to(synth.make_pretrain_batch(B=4, T=[2, 3, 1, 2], L=[30, 22, 16, 25] if rank == 0 else [27, 19, 12, 21], seed=50 + rank,
                                              vocab_size=1000, style='rich'), 'cuda')
w = dp.GoatDataParallel(model)
tasks = ('mlm', 'sap', 'cfp')
for t in tasks:
    for p in model.parameters():
        p.grad = None
    model(gb, t, True).mean().backward()
    w.record_usage(t)
for p in model.parameters():
    p.grad = None
arena = w.build_arena(late_prefixes=('bert.embeddings.', 'bert.lang_encoder.'))
box = {}
def mark(m, i, o):
    v = o.view_as(o); box['txt'] = v; return v
model.bert.lang_encoder.register_forward_hook(mark)

def dense(task):
    arena.zero(task)
    model(gb, task, True).mean().backward()
    w.reduce_gradients(task)
    torch.cuda.synchronize()
    return arena.flat.clone()
ref = {t: dense(t) for t in tasks}
w.enable_sparse_embedding(model.bert.embeddings.word_embeddings.weight, ['sap', 'cfp'], mixed_tasks=['mlm'])   # mlm: dense decoder part early, lookups sparse
ok = True
assert w._sparse[2] == {'mlm'}, 'mlm must take the mixed dense + sparse path in this check'
for t in tasks * 2:
    arena.zero(t)
    w.begin_step(t)
    loss = model(gb, t, True)
    w.backward_phase_a(loss.mean(), box['txt'])
    w.reduce_gradients(t, phase=0, wait=False)
    w.backward_phase_b(box['txt'])
    w.reduce_gradients(t, phase=1)
    torch.cuda.synchronize()
    assert len(w._stash[t]) == 1, 'the lookup gradient of the word table must have gone through the sparse exchange'
    got = arena.flat
    worst = 0.0
    for p in arena.params:
        tset = arena.tasks_of[id(p)]
        if tset is not None and t not in tset:
            continue
        a = arena.offsets[id(p)]
        g, r = got[a:a + p.numel()].double(), ref[t][a:a + p.numel()].double()
        worst = max(worst, float((g - r).norm() / max(float(r.norm()), 1e-3 * float(ref[t].double().norm()))))
    if rank == 0:
        print('task %s: worst relative deviation from the dense single-backward path %.2e' % (t, worst))
    ok = ok and worst < 5e-3
flag = torch.tensor([int(ok)]); dist.all_reduce(flag)
if rank == 0:
    print('DP_SPARSE_CHECK_OK' if int(flag) == world else 'DP_SPARSE_CHECK_FAILED')
dist.barrier(); dist.destroy_process_group()
