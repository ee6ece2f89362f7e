"""SURVEY §8 a-18 on the device: the pre-training step harness (vln_goat_amd.train_step.PretrainStep) drives the HIP model
with the gradient arena attached — forward, backward, clip_grad_norm_(5.0), AdamW — and the loss of a fixed batch falls."""
import pytest
import torch


def _goat_graph(g, **kw):
    """torch.cuda.graph through vln_goat_amd.hipops.graph: a graph whose capture forked one of the package's parallel branches is kept
    alive (ROCm 7.2 graph-destruction bug; see hipops.graph)."""
    from vln_goat_amd import hipops
    return hipops.graph(g, **kw)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('use_arena', [False, True])
def test_pretrain_step_trains_the_hip_model(use_arena):
    import vln_goat_amd
    from vln_goat_amd import config as gcfg, dp, pretrain_model, synth, train_step
    cfg = gcfg.make_config(num_l_layers=2, num_top_layer=2, num_pano_layers=1, vocab_size=1000,
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = pretrain_model.GlocalTextPathCMTPreTraining(cfg).cuda().train()
    gb = synth.batch_to(synth.make_pretrain_batch(B=4, T=[2, 3, 1, 2], L=[30, 22, 16, 25], seed=5, vocab_size=1000, style='rich'), 'cuda')
    vln_goat_amd.set_compute_dtype(torch.bfloat16)
    try:
        wrapper = dp.GoatDataParallel(model)
        if use_arena:
            for t in ('mlm', 'sap', 'cfp'):
                for p in model.parameters():
                    p.grad = None
                model(gb, t, True).mean().backward()
                wrapper.record_usage(t)
            for p in model.parameters():
                p.grad = None
            wrapper.build_arena()
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=2e-4, weight_decay=0.01)
        step = train_step.PretrainStep(model, opt, grad_accum=1, grad_norm=5.0, wrapper=wrapper)
        first, last = {}, {}
        for it in range(8):
            for name in ('mlm', 'sap_r2r', 'cfp'):
                info = step(name, gb)
                assert info['updated'] and info['task'] == name.split('_')[0]
                assert info['grad_norm'] is not None and info['grad_norm'] == info['grad_norm']      # finite, not NaN
                assert info['n_loss_units'] > 0
                first.setdefault(name, info['loss'])
                last[name] = info['loss']
        torch.cuda.synchronize()
        assert step.global_step == 24
        for name in first:
            assert last[name] < first[name], (name, first[name], last[name])
    finally:
        vln_goat_amd.set_compute_dtype(torch.float32)


def _total_norm(fp):
    import numpy as np
    return float(np.sqrt((fp[:, 0].astype(np.float64) ** 2).sum()))


@pytest.mark.parametrize('use_arena', [False, True])
def test_pretrain_step_matches_config1_golden(use_arena):
    """the harness driven against the reference capture it is pinned to — config 1 (2/2/2 layers, full
    vocabulary, B=4), dropout off: per task the mean of the reference's loss vector, the clipped-gradient norm
    (clip_grad_norm_ 5.0, P/train_r2r_goat.py:349-354) and the L2 norm + leading elements of every parameter's .grad as the
    optimizer sees it (after clipping), fp32 <= 1e-3.  lr = 0: the weights stay at the golden state for every task."""
    import numpy as np
    import vln_goat_amd
    from helpers import build_case, case_tasks, fingerprint, load_golden
    from vln_goat_amd import dp, synth, train_step
    case = 'pretrain_config1'
    cfg, model, batch = build_case(case)
    gold = load_golden(case)
    model = model.cuda().eval()
    gb = synth.batch_to(batch, 'cuda')
    wrapper = dp.GoatDataParallel(model)
    if use_arena:
        for t in case_tasks(case):
            for p in model.parameters():
                p.grad = None
            model(gb, t, True).mean().backward()
            wrapper.record_usage(t)
        for p in model.parameters():
            p.grad = None
        wrapper.build_arena()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.0)
    step = train_step.PretrainStep(model, opt, grad_accum=1, grad_norm=5.0, wrapper=wrapper)
    names = [str(n) for n in gold['param_names']]
    params = dict(model.named_parameters())
    for rnd in range(2):                                     # twice: the second round runs on the arena's learned owner sets
        for task in case_tasks(case):
            info = step(task, gb)
            torch.cuda.synchronize()
            ref_loss = float(gold[task + '_loss_vec'].mean())
            assert abs(info['loss'] - ref_loss) < 1e-3 * max(1.0, abs(ref_loss)), (task, info['loss'], ref_loss)
            assert info['n_loss_units'] == gold[task + '_loss_vec'].shape[0]
            fp = gold[task + '_grad_fp']
            tot = _total_norm(fp)
            assert abs(info['grad_norm'] - tot) < 2e-3 * tot, (task, info['grad_norm'], tot)
            coef = min(1.0, 5.0 / (tot + 1e-6))
            gmax = float(fp[:, 0].max()) * coef
            for i, n in enumerate(names):
                ref, g = fp[i] * coef, params[n].grad
                if fp[i][0] == 0.0:                          # the reference left .grad at None (parameter unused by the task)
                    assert g is None or float(g.norm()) <= 1e-5 * gmax, (task, n)
                    continue
                got = fingerprint(g)
                assert np.abs(got - ref).max() <= 2e-3 * max(ref[0], 1e-4 * gmax), (task, n, got, ref)


def test_arena_step_updates_only_the_parameters_of_the_task():
    """with the gradient arena attached a sap / cfp step must leave the parameters only mlm uses (mlm_head.*)
    exactly where they are — their arena slices still hold the last mlm gradient, and an optimizer that saw it would apply it
    again (moments and weight decay included).  The arena run must match the arena-less run weight for weight."""
    import vln_goat_amd
    from vln_goat_amd import config as gcfg, dp, pretrain_model, synth, train_step
    cfg = gcfg.make_config(num_l_layers=1, num_top_layer=1, num_pano_layers=1, vocab_size=500,
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    gb = synth.batch_to(synth.make_pretrain_batch(B=3, T=[2, 1, 2], L=[20, 14, 9], seed=4, vocab_size=500, style='rich'), 'cuda')
    runs = []
    for use_arena in (False, True):
        torch.manual_seed(0)
        model = pretrain_model.GlocalTextPathCMTPreTraining(cfg).cuda().train()
        wrapper = dp.GoatDataParallel(model)
        if use_arena:
            for t in ('mlm', 'sap', 'cfp'):
                for p in model.parameters():
                    p.grad = None
                model(gb, t, True).mean().backward()
                wrapper.record_usage(t)
            for p in model.parameters():
                p.grad = None
            wrapper.build_arena()
        # (SGD with momentum and weight decay: linear in the gradient, so mathematically-zero gradients — rounding noise of
        # atomically ordered sums — stay noise instead of being normalised to +-lr as Adam would)
        opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.05, momentum=0.9, weight_decay=0.05)
        step = train_step.PretrainStep(model, opt, grad_accum=1, grad_norm=5.0, wrapper=wrapper)
        snaps = []
        for name in ('mlm', 'sap', 'cfp', 'sap', 'mlm'):
            step(name, gb)
            torch.cuda.synchronize()
            snaps.append({n: p.detach().clone() for n, p in model.named_parameters()})
        runs.append(snaps)
    plain, arena = runs
    for k, (a, b) in enumerate(zip(plain, arena)):
        for n in a:
            assert torch.allclose(a[n], b[n], rtol=2e-4, atol=2e-6), (k, n, float((a[n] - b[n]).abs().max()))
    head = [n for n in plain[0] if n.startswith('mlm_head.predictions.transform')]
    assert head
    for n in head:                                               # untouched by the sap and cfp steps that follow the first mlm step
        assert torch.equal(arena[0][n], arena[1][n]) and torch.equal(arena[1][n], arena[2][n]) and torch.equal(arena[2][n], arena[3][n]), n
        assert not torch.equal(arena[3][n], arena[4][n]), n      # and moved again by the second mlm step


def test_fused_adamw_matches_reference_trace():
    """N3: goat_grad_sqnorm + goat_adamw_step on the gradient arena against three steps of the REFERENCE optimizer
    (P/optim/adamw.py:53-110 behind clip_grad_norm_(5.0); tests/golden/make_contract.py wrote the trace): a clipped step, two
    unclipped ones, a parameter that never gets a gradient, one that skips a step (its bias correction lags), a changing
    learning rate, weight decay on matrices only, bf16 shadows refreshed in the same pass."""
    import numpy as np
    from helpers import load_golden
    from vln_goat_amd import dp, hipops, optim
    tr = load_golden('adamw_trace')
    names = [str(n) for n in tr['names']]
    params = {n: torch.nn.Parameter(torch.from_numpy(tr['p0_' + n]).cuda()) for n in names}
    usage = {id(params[n]): ({'a', 'b'} if n not in ('head.weight', 'unused.weight') else ({'a'} if n == 'head.weight' else set())) for n in names}
    arena = dp.GradArena(list(params.values()), usage).attach()
    assert id(params['unused.weight']) not in arena.views
    opt = optim.FusedAdamW(params.items(), arena, lr=5e-5, betas=(0.9, 0.98), weight_decay=0.01)
    assert sorted(opt.param_groups[1]['names']) == ['enc.LayerNorm.bias', 'enc.LayerNorm.weight', 'enc.dense.bias']
    sh = hipops._shadow(params['enc.dense.weight'], torch.bfloat16)                  # a cached bf16 operand copy
    for step, task in enumerate(('a', 'b', 'a')):
        arena.bind(task)
        arena.flat.zero_()
        for n in names:
            if ('g%d_%s' % (step, n)) in tr:
                arena.views[id(params[n])].copy_(torch.from_numpy(tr['g%d_%s' % (step, n)]))
        for g in opt.param_groups:
            g['lr'] = float(tr['lr'][step])
        opt.step(task, max_norm=5.0)
        torch.cuda.synchronize()
        assert abs(opt.last_grad_norm() - float(tr['gnorm%d' % step][0])) < 1e-4 * float(tr['gnorm%d' % step][0])
        for n in names:
            ref = tr['p%d_%s' % (step + 1, n)]
            got = params[n].detach().cpu().numpy()
            assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()) + 1e-9, (step, n, np.abs(got - ref).max())
        assert torch.equal(sh, params['enc.dense.weight'].detach().to(torch.bfloat16))       # refreshed in the same pass
        assert hipops._shadow(params['enc.dense.weight'], torch.bfloat16) is sh               # and still the cached object
    for n in names:
        if ('m_' + n) in tr:
            stt = opt.state_of(params[n])
            assert np.abs(stt['exp_avg'].cpu().numpy() - tr['m_' + n]).max() <= 1e-6 * max(1e-3, np.abs(tr['m_' + n]).max())
            assert np.abs(stt['exp_avg_sq'].cpu().numpy() - tr['v_' + n]).max() <= 3e-6 * max(1e-6, np.abs(tr['v_' + n]).max())
    assert opt.steps[id(params['head.weight'])] == 2 and opt.steps[id(params['emb.word.weight'])] == 3


def _arena_for(model, gb, tasks=('mlm', 'sap', 'cfp')):
    from vln_goat_amd import dp
    wrapper = dp.GoatDataParallel(model)
    for t in tasks:
        for p in model.parameters():
            p.grad = None
        model(gb, t, True).mean().backward()
        wrapper.record_usage(t)
    for p in model.parameters():
        p.grad = None
    return wrapper, wrapper.build_arena()


def test_fused_adamw_on_the_model_matches_torch_adamw():
    """The fused step inside PretrainStep on the real model (fp32 path: deterministic operands) next to torch.optim.AdamW with the
    reference's two parameter groups: same weights after two rounds of mlm / sap / cfp.  (torch applies the decoupled decay
    before the update and the reference after it, and they place eps differently: second-order differences, far below the
    tolerance; parameters whose gradient is rounding noise are normalised to +-lr by any Adam and are left out.)"""
    from vln_goat_amd import config as gcfg, optim, pretrain_model, synth, train_step
    cfg = gcfg.make_config(num_l_layers=1, num_top_layer=1, num_pano_layers=1, vocab_size=400,
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    gb = synth.batch_to(synth.make_pretrain_batch(B=3, T=[2, 1, 2], L=[20, 14, 9], seed=4, vocab_size=400, style='rich'), 'cuda')
    finals, gnorms = [], []
    for fused in (False, True):
        torch.manual_seed(0)
        model = pretrain_model.GlocalTextPathCMTPreTraining(cfg).cuda().train()
        wrapper, arena = _arena_for(model, gb)
        if fused:
            opt = optim.FusedAdamW(model.named_parameters(), arena, lr=1e-4, betas=(0.9, 0.98), eps=1e-10, weight_decay=0.01)
        else:
            named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
            nd = [p for n, p in named if any(k in n for k in optim.NO_DECAY)]
            dc = [p for n, p in named if not any(k in n for k in optim.NO_DECAY)]
            opt = torch.optim.AdamW([{'params': dc, 'weight_decay': 0.01}, {'params': nd, 'weight_decay': 0.0}], lr=1e-4, betas=(0.9, 0.98), eps=1e-10)
        step = train_step.PretrainStep(model, opt, grad_accum=1, grad_norm=5.0, wrapper=wrapper)
        start = {n: p.detach().clone() for n, p in model.named_parameters()}
        noisy = set()
        for rnd in range(2):
            for name in ('mlm', 'sap', 'cfp'):
                info = step(name, gb)
                gn = info['grad_norm']
                gnorms.append(gn() if callable(gn) else gn)
                if not fused:
                    gmax = max(float(p.grad.norm()) for p in model.parameters() if p.grad is not None)
                    noisy |= {n for n, p in model.named_parameters() if p.grad is not None and float(p.grad.norm()) < 1e-5 * gmax}
        torch.cuda.synchronize()
        finals.append(({n: p.detach().clone() for n, p in model.named_parameters()}, start, noisy))
    (a, start, noisy), (b, _, _) = finals
    assert all(abs(x - y) < 1e-3 * x for x, y in zip(gnorms[:6], gnorms[6:])), gnorms
    moved = 0
    for n in a:
        if n in noisy or 'key.bias' in n or 'in_proj_bias' in n:      # (key biases: softmax is shift-invariant, their gradient is rounding noise)
            continue
        da, db = (a[n] - start[n]).double(), (b[n] - start[n]).double()
        if float(da.norm()) == 0.0:
            assert float(db.norm()) == 0.0, n                     # parameters no task uses are never touched
            continue
        moved += 1
        assert float((da - db).norm()) <= 2e-2 * float(da.norm()) + 1e-9, (n, float((da - db).norm()), float(da.norm()))
    assert moved > 100


def test_fused_adamw_keeps_the_bf16_shadows_coherent():
    """bf16 path: the fused step refreshes the operand shadows (plain, row-concatenated QKV / KV, vocabulary-padded decoder) in
    its own pass.  After a few steps every cached bf16 copy equals the rounded float32 master, and a forward pass on the cached
    copies equals the forward pass after dropping every cache (copies rebuilt from the masters)."""
    import vln_goat_amd
    from vln_goat_amd import config as gcfg, optim, pretrain_model, synth, train_step
    cfg = gcfg.make_config(num_l_layers=2, num_top_layer=1, num_pano_layers=1, vocab_size=1000,
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = pretrain_model.GlocalTextPathCMTPreTraining(cfg).cuda().train()
    gb = synth.batch_to(synth.make_pretrain_batch(B=4, T=[2, 3, 1, 2], L=[30, 22, 16, 25], seed=5, vocab_size=1000, style='rich'), 'cuda')
    vln_goat_amd.set_compute_dtype(torch.bfloat16)
    try:
        wrapper, arena = _arena_for(model, gb)
        opt = optim.FusedAdamW(model.named_parameters(), arena, lr=2e-4, betas=(0.9, 0.98), weight_decay=0.01)
        step = train_step.PretrainStep(model, opt, grad_accum=1, grad_norm=5.0, wrapper=wrapper)
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        for it in range(2):
            for name in ('mlm', 'sap', 'cfp'):
                assert step(name, gb)['updated']
        torch.cuda.synchronize()
        assert sum(1 for n, p in model.named_parameters() if not torch.equal(p.detach(), before[n])) > 100
        checked = {'plain': 0, 'cat': 0, 'rowpad': 0, 'catb': 0, 'kpad': 0}
        assert opt.last_refreshed == 0          # every cached copy of the bf16 state is refreshed by the update kernel itself
        byid = {id(q): q for q in model.parameters()}
        for n, p in model.named_parameters():
            for k, (ver, t) in (p.__dict__.get('_goat_shadow') or {}).items():
                if k[0] == 'cat' and t.dtype == torch.bfloat16:
                    assert torch.equal(t, torch.cat([byid[i].detach() for i in k[3]], 0).to(torch.bfloat16)), n
                    checked['cat'] += 1
                elif k[0] == 'rowpad':
                    assert torch.equal(t[:p.shape[0]], p.detach().to(torch.bfloat16)) and not bool(t[p.shape[0]:].any()), n
                    checked['rowpad'] += 1
                elif k[0] == 'catb':
                    assert torch.equal(t, torch.cat([byid[i].detach().float() for i in k[1]], 0)), n
                    checked['catb'] += 1
                elif k[0] == torch.bfloat16 and k[1] is False and k[2] == 0:
                    assert torch.equal(t, p.detach().to(torch.bfloat16)), n
                    checked['plain'] += 1
                elif k[0] == torch.bfloat16 and k[1] is False and k[2] > 0:       # K-padded copies of the 7- / 14-wide position Linears
                    assert torch.equal(t[:, :p.shape[1]], p.detach().to(torch.bfloat16)) and not bool(t[:, p.shape[1]:].any()), n
                    checked['kpad'] += 1
        assert checked['plain'] > 10 and checked['cat'] > 3 and checked['rowpad'] == 1 and checked['catb'] > 3 and checked['kpad'] >= 2, checked
        with torch.no_grad():
            cached = [model(gb, t, True).float().clone() for t in ('mlm', 'sap', 'cfp')]
            for p in model.parameters():
                p.__dict__.pop('_goat_shadow', None)
            fresh = [model(gb, t, True).float() for t in ('mlm', 'sap', 'cfp')]
        for x, y in zip(cached, fresh):
            assert torch.equal(x, y)
    finally:
        vln_goat_amd.set_compute_dtype(torch.float32)


def test_captured_step_replayed_after_fused_adamw_reads_the_updated_weights():
    """A hipGraph has the addresses of the cached operand copies baked in (bf16 shadows, concatenated QKV weights AND biases, K-padded
    position Linears).  FusedAdamW writes the masters through raw pointers and must refresh every copy at its address: after
    opt.step() the replayed graph has to return what the eager model returns on the updated weights (the
    concatenated biases and padded copies used to be dropped, so replays kept reading freed / stale memory)."""
    import vln_goat_amd
    from vln_goat_amd import config as gcfg, hipops, optim, pretrain_model, synth
    cfg = gcfg.make_config(num_l_layers=2, num_top_layer=1, num_pano_layers=1, vocab_size=1000,
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = pretrain_model.GlocalTextPathCMTPreTraining(cfg).cuda().train()
    gb = synth.batch_to(synth.make_pretrain_batch(B=4, T=[2, 3, 1, 2], L=[30, 22, 16, 25], seed=5, vocab_size=1000, style='rich'), 'cuda')
    vln_goat_amd.set_compute_dtype(torch.bfloat16)
    try:
        wrapper, arena = _arena_for(model, gb)
        # a large step so that stale copies would be visible far above the bf16 noise
        opt = optim.FusedAdamW(model.named_parameters(), arena, lr=2e-3, betas=(0.9, 0.98), weight_decay=0.01)
        out = {}

        def body(task):
            arena.zero(task)
            loss = model(gb, task, compute_loss=True)
            loss.mean().backward()
            out[task] = loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for t in ('mlm', 'sap', 'cfp'):
                body(t)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        out.clear()          # (the warm-up's autograd graphs — AccumulateGrad nodes bound to the side stream — must not outlive into the capture)
        graphs = {}
        for t in ('mlm', 'sap', 'cfp'):
            g = torch.cuda.CUDAGraph()
            with _goat_graph(g):
                body(t)
            graphs[t] = g
        for rnd in range(2):
            for t in ('mlm', 'sap', 'cfp'):
                graphs[t].replay()
                opt.step(t, max_norm=5.0)
        torch.cuda.synchronize()
        biases = [p for n, p in model.named_parameters() if n.endswith('query.bias') and id(p) in arena.views]
        assert len(biases) > 4 and all(float(b.detach().abs().max()) > 1e-3 for b in biases)      # the QKV biases (zero at init) did move
        for t in ('mlm', 'sap', 'cfp'):
            graphs[t].replay()
            torch.cuda.synchronize()
            got = out[t].detach().float().clone()
            got_grad = arena.flat.detach().clone()
            by_id = {id(p): p for p in model.parameters()}     # eager on copies rebuilt (in place) from the float32 masters
            for p in model.parameters():
                hipops.refresh_shadows(p, by_id)
            arena.zero(t)
            ref = model(gb, t, compute_loss=True)
            ref.mean().backward()
            torch.cuda.synchronize()
            assert torch.allclose(got, ref.detach().float(), rtol=2e-2, atol=2e-2), (t, got, ref)
            scale = float(arena.flat.abs().max())
            assert float((got_grad - arena.flat).abs().max()) <= 3e-2 * scale, t
    finally:
        vln_goat_amd.set_compute_dtype(torch.float32)


def test_static_batch_feeds_a_captured_step_with_new_batches():
    """train_step.StaticBatch: the hipGraph captured on the fixed-address batch, replayed after pack / stage / commit of a
    NEW host batch (other ids, lengths, features, map strings), returns the losses and gradients the eager model computes on
    that batch — index tensors and memoised masks follow the new data (dropout off: deterministic)."""
    import vln_goat_amd
    from vln_goat_amd import config as gcfg, hipops, pretrain_model, synth, train_step
    cfg = gcfg.make_config(num_l_layers=2, num_top_layer=2, num_pano_layers=1, vocab_size=1000,
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = pretrain_model.GlocalTextPathCMTPreTraining(cfg).cuda().eval()
    mk = lambda seed: synth.make_pretrain_batch(B=4, T=3, L=30, seed=seed, vocab_size=1000, style='survey')
    first, second = mk(5), mk(6)
    second['txt_lens'] = torch.tensor([30, 11, 23, 17])            # other text lengths: the key masks must follow
    vln_goat_amd.set_compute_dtype(torch.bfloat16)
    try:
        sb = train_step.StaticBatch(cfg, first)
        params = [p for p in model.parameters() if p.requires_grad]
        out = {}

        def step(task):
            for p in params:
                p.grad = None
            loss = model(sb.gb, task, compute_loss=True)
            loss.mean().backward()
            out[task] = loss

        graphs = {}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for t in ('mlm', 'sap', 'cfp'):
                step(t)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        out.clear()          # (no warm-up autograd graph alive during the capture)
        grads = {}
        for t in ('mlm', 'sap', 'cfp'):
            g = torch.cuda.CUDAGraph()
            with _goat_graph(g):
                step(t)
            graphs[t] = g
            grads[t] = {id(p): p.grad for p in params if p.grad is not None}
        buf = sb.pack(second)
        ev = sb.stage(buf)
        sb.commit()
        assert ev is not None
        for t in ('mlm', 'sap', 'cfp'):
            graphs[t].replay()
            torch.cuda.synchronize()
            got_loss = out[t].detach().float().clone()
            got = {k: v.detach().float().clone() for k, v in grads[t].items()}
            gb = synth.batch_to(second, 'cuda')                   # eager on a plain device copy of the same host batch
            for p in params:
                p.grad = None
            ref = model(gb, t, compute_loss=True)
            ref.mean().backward()
            torch.cuda.synchronize()
            assert torch.allclose(got_loss, ref.detach().float(), rtol=1e-3, atol=1e-3), (t, got_loss, ref)
            n = 0
            top = max(float(p.grad.abs().max()) for p in params if p.grad is not None)
            for p in params:
                if p.grad is None:
                    continue
                a, b = got[id(p)], p.grad.float()
                # (per tensor, with a floor: key biases have an identically zero gradient — softmax shift invariance — and
                #  hold rounding noise only)
                scale = max(float(b.abs().max()), 0.05 * top)
                assert float((a - b).abs().max()) <= 2e-2 * scale, t
                n += 1
            assert n > 10
        with pytest.raises(ValueError):
            sb.pack(synth.make_pretrain_batch(B=4, T=2, L=30, seed=7, vocab_size=1000, style='survey'))
    finally:
        vln_goat_amd.set_compute_dtype(torch.float32)


def test_shape_bucketed_static_batch_replays_ragged_batches():
    """StaticBatch(bucket=...): ONE captured step per task serves RAGGED batches (other trajectory lengths, text lengths, map sizes,
    bf16 features from a features.FeatureStore-style host table) padded into the bucket — loss mean and every gradient equal the
    eager model on the UNPADDED batch (dropout off).  Covers the padded MLM selection (ignored rows + mean rescale), the fused
    panorama rows behind padded view rows, and the CFP pooling masks that keep the reference's pooling width."""
    import vln_goat_amd
    from vln_goat_amd import config as gcfg, pretrain_model, synth, train_step
    cfg = gcfg.make_config(num_l_layers=2, num_top_layer=2, num_pano_layers=1, vocab_size=1000,
                           hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = pretrain_model.GlocalTextPathCMTPreTraining(cfg).cuda().eval()

    def mk(seed, T, L, **kw):
        b = synth.make_pretrain_batch(B=4, T=T, L=L, seed=seed, vocab_size=1000, style='rich', **kw)
        b['traj_view_img_fts'] = b['traj_view_img_fts'].to(torch.bfloat16)          # the bf16 feature store's rows
        return b
    # `second` has ragged panoramas whose LAST steps all hold fewer than 36 views: its own local width is 32 (31 views + [stop]) inside the
    # bucket's 37 — the CFP pooling of the local branch must stay on those 32 slots (cfp_vp_mask)
    first, second, third = mk(5, [3, 2, 4, 3], [30, 22, 16, 25]), mk(20, [1, 4, 2, 2], [12, 28, 9, 20], ragged_views=True), mk(7, [4, 4, 3, 4], [32, 32, 5, 17])
    assert second['vp_pos_fts'].shape[1] == 32 and first['vp_pos_fts'].shape[1] == 37 and second['traj_view_img_fts'].shape[1] == 36
    G = max(b['gmap_step_ids'].shape[1] for b in (first, second, third))
    vln_goat_amd.set_compute_dtype(torch.bfloat16)
    try:
        sb = train_step.StaticBatch(cfg, first, bucket=dict(L=32, N=16, G=G + 2))
        params = [p for p in model.parameters() if p.requires_grad]
        out = {}

        def step(task):
            for p in params:
                p.grad = None
            loss = model(sb.gb, task, compute_loss=True)
            loss.mean().backward()
            out[task] = loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for t in ('mlm', 'sap', 'cfp'):
                step(t)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        out.clear()          # (see above: no warm-up autograd graph alive during the capture)
        graphs, grads = {}, {}
        for t in ('mlm', 'sap', 'cfp'):
            g = torch.cuda.CUDAGraph()
            with _goat_graph(g):
                step(t)
            graphs[t] = g
            grads[t] = {id(p): p.grad for p in params if p.grad is not None}
        for host in (second, third, first):
            assert sb.fits(host)
            sb.stage(sb.pack(host))
            sb.commit()
            for t in ('mlm', 'sap', 'cfp'):
                graphs[t].replay()
                torch.cuda.synchronize()
                got_mean = float(out[t].detach().float().mean())
                got = {k: v.detach().float().clone() for k, v in grads[t].items()}
                gb = synth.batch_to(host, 'cuda')                      # eager on the unpadded batch
                for p in params:
                    p.grad = None
                ref = model(gb, t, compute_loss=True)
                ref.mean().backward()
                torch.cuda.synchronize()
                ref_mean = float(ref.detach().float().mean())
                assert abs(got_mean - ref_mean) <= 2e-2 * max(1.0, abs(ref_mean)), (t, got_mean, ref_mean)
                top = max(float(p.grad.abs().max()) for p in params if p.grad is not None)
                n = 0
                for p in params:
                    if p.grad is None:
                        continue
                    a, b = got[id(p)], p.grad.float()
                    scale = max(float(b.abs().max()), 0.05 * top)
                    assert float((a - b).abs().max()) <= 3e-2 * scale, (t, n)
                    n += 1
                assert n > 10
    finally:
        vln_goat_amd.set_compute_dtype(torch.float32)


def test_captured_steps_launch_only_measured_configurations():
    """round 5's largest gain was a silent plan miss (the captured steps ran untuned weight-gradient groups because the eager
    warm-up formed other groups than the capture).  Procedure of bench.py / a trainer: warm up each task with the tuner on and the parallel
    branches forked as a capture forks them (Branch.like_capture), then capture.  The capture must not contain ONE GEMM on the static
    heuristic or ONE weight-gradient group on the default tile (tuning.STATS counts them; STATS_LOG names them).  The same capture after
    a warm-up WITHOUT like_capture is the negative control: it must show the misses (otherwise the counters count nothing)."""
    import vln_goat_amd
    from vln_goat_amd import config as gcfg, hipops, pretrain_model, synth, tuning
    cfg = gcfg.make_config(num_l_layers=2, num_top_layer=1, num_pano_layers=1, vocab_size=1000)
    torch.manual_seed(0)
    model = pretrain_model.GlocalTextPathCMTPreTraining(cfg).cuda().train()
    gb = synth.batch_to(synth.make_pretrain_batch(B=6, T=[2, 3, 1, 2, 4, 3], L=[30, 22, 16, 25, 40, 33], seed=5, vocab_size=1000, style='rich'), 'cuda')
    vln_goat_amd.set_compute_dtype(torch.bfloat16)
    saved = (dict(tuning._TUNED), dict(hipops.WgradQueue.tuned), tuning.AUTOTUNE, hipops.WgradQueue.FORCE_TUNE)
    try:
        wrapper, arena = _arena_for(model, gb)
        hipops.RngState.dev = torch.zeros(1, dtype=torch.int64, device='cuda')

        def body(task):
            arena.zero(task)
            hipops.RngState.dev.add_(1)
            model(gb, task, compute_loss=True).mean().backward()

        def warm(like_capture):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ctx = hipops.Branch.like_capture() if like_capture else __import__('contextlib').nullcontext()
                with ctx:
                    for t in ('mlm', 'sap', 'cfp'):
                        body(t)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()

        def capture_all():
            tuning.reset_stats()
            graphs = []
            for t in ('mlm', 'sap', 'cfp'):
                g = torch.cuda.CUDAGraph()
                with _goat_graph(g):
                    body(t)
                graphs.append(g)
            torch.cuda.synchronize()
            return graphs, dict(tuning.STATS), list(tuning.STATS_LOG)

        # negative control first (tables empty, tuner on, warm-up on ONE stream): the capture's weight-gradient groups were never timed
        tuning._TUNED.clear()
        hipops.WgradQueue.tuned.clear()
        tuning.AUTOTUNE, hipops.WgradQueue.FORCE_TUNE = True, True
        warm(like_capture=False)
        _, st0, log0 = capture_all()
        assert st0['gemm_heuristic'] == 0, log0[:5]                       # GEMM shapes do not depend on the stream they are issued on
        assert st0['wgrad_default'] > 0, st0                              # the groups do: this is the round-5 bug, visible to the counter
        # the procedure: warm up as the capture will run
        warm(like_capture=True)
        graphs, st1, log1 = capture_all()
        assert st1['gemm_heuristic'] == 0 and st1['wgrad_default'] == 0, (st1, log1[:8])
        assert st1['gemm_tuned'] > 30 and st1['wgrad_tuned'] >= 3, st1
        for g in graphs:        # and the captured steps replay
            g.replay()
        torch.cuda.synchronize()
        assert torch.isfinite(arena.flat).all()
    finally:
        tuning._TUNED.clear(); tuning._TUNED.update(saved[0])
        hipops.WgradQueue.tuned.clear(); hipops.WgradQueue.tuned.update(saved[1])
        tuning.AUTOTUNE, hipops.WgradQueue.FORCE_TUNE = saved[2], saved[3]
        vln_goat_amd.set_compute_dtype(torch.float32)
