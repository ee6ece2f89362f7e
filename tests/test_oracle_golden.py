"""The CPU oracle (oracle/goat_oracle.py) against golden vectors produced by the imported reference
(tests/golden/make_golden_pretrain.py).  This pins the oracle: <= 1e-5 on losses/logits, gradients of
every parameter to 1e-4 relative on the L2 norm and the leading elements."""
import os

import numpy as np
import pytest
import torch

from helpers import CASES, build_case, case_tasks, check_projections, fingerprint, load_golden, oracle_run, projections, ROOT

SMALL = ['pretrain_small_fixed', 'pretrain_small_ragged']
ALL = SMALL + (['pretrain_config1'] if os.path.exists(os.path.join(ROOT, 'tests/golden/pretrain_config1.npz')) else [])


EXTRA = ['pretrain_reverie_small', 'pretrain_r2r_mrc']      # REVERIE object branch + OG head, MRC head
BACL = ['pretrain_bacl_type2_door', 'pretrain_bacl_type1_xattn']      # BACL-txt in pre-training (do_back_txt)
CASE_TASKS = [(c, t) for c in ALL + EXTRA + BACL for t in case_tasks(c)]


@pytest.mark.parametrize('case,task', CASE_TASKS)
def test_oracle_matches_reference_golden(case, task):
    gold = load_golden(case)
    cfg, model, batch = build_case(case)
    sd = model.state_dict()
    loss_vec, grads = oracle_run(cfg, sd, batch, task)
    np.testing.assert_allclose(loss_vec.numpy(), gold[task + '_loss_vec'], rtol=1e-5, atol=1e-5)
    names = [str(n) for n in gold['param_names']]
    fp = gold[task + '_grad_fp']
    worst = 0.0
    for i, n in enumerate(names):
        got = fingerprint(grads.get(n))
        ref = fp[i]
        scale = max(abs(ref[0]), 1e-6)
        err = np.abs(got - ref).max() / scale
        worst = max(worst, err)
        assert err < 2e-4, (n, got, ref)
    assert worst < 2e-4
    # seeded random projections of EVERY element of every gradient tensor (the fingerprint only sees the norm and 8 elements)
    proj = gold[task + '_grad_proj']
    gmax = float(fp[:, 0].max())
    for i, n in enumerate(names):
        check_projections(projections(grads.get(n)), proj[i], max(float(fp[i][0]), 1e-3 * gmax), 2e-4, n)


@pytest.mark.parametrize('case', SMALL)
def test_oracle_logits_and_pooled_vectors(case):
    from oracle import goat_oracle
    gold = load_golden(case)
    cfg, model, batch = build_case(case)
    sd = model.state_dict()
    with torch.no_grad():
        gl, ll, fl = goat_oracle.forward(cfg, sd, batch, 'sap', compute_loss=False)
        go, vo, fo, to = goat_oracle.forward(cfg, sd, batch, 'cfp', compute_loss=False)
        sc = goat_oracle.forward(cfg, sd, batch, 'mlm', compute_loss=False)
    for got, key in ((gl, 'sap_global_logits'), (ll, 'sap_local_logits'), (fl, 'sap_fused_logits')):
        ref = gold[key]
        assert np.array_equal(np.isinf(got.numpy()), np.isinf(ref))
        m = ~np.isinf(ref)
        np.testing.assert_allclose(got.numpy()[m], ref[m], rtol=1e-5, atol=2e-5)
    for got, key in ((go, 'cfp_gmap_out'), (vo, 'cfp_vp_out'), (fo, 'cfp_fused_out'), (to, 'cfp_txt_out')):
        np.testing.assert_allclose(got.numpy(), gold[key], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(sc[:, :64].numpy(), gold['mlm_scores_head'], rtol=1e-5, atol=5e-5)
    np.testing.assert_allclose(torch.logsumexp(sc, 1).numpy(), gold['mlm_scores_lse'], rtol=1e-5, atol=5e-5)


@pytest.mark.parametrize('case', EXTRA)
def test_oracle_og_and_mrc_outputs(case):
    from oracle import goat_oracle
    gold = load_golden(case)
    cfg, model, batch = build_case(case)
    sd = model.state_dict()
    with torch.no_grad():
        vp, vt, op, ot = goat_oracle.forward(cfg, sd, batch, 'mrc', compute_loss=False)
        np.testing.assert_allclose(vp.numpy(), gold['mrc_view_pred'], rtol=1e-5, atol=5e-5)
        if 'mrc_obj_pred' in gold:
            np.testing.assert_allclose(op.numpy(), gold['mrc_obj_pred'], rtol=1e-5, atol=5e-5)
        else:
            assert op is None
        if 'og' in case_tasks(case):
            lg = goat_oracle.forward(cfg, sd, batch, 'og', compute_loss=False).numpy()
            ref = gold['og_logits']
            assert np.array_equal(np.isinf(lg), np.isinf(ref))
            np.testing.assert_allclose(lg[~np.isinf(ref)], ref[~np.isinf(ref)], rtol=1e-5, atol=2e-5)
