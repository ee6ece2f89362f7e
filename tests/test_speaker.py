"""SURVEY §8f N4 (last item): the speaker (vln_goat_amd.speaker.Transpeaker) against the IMPORTED REFERENCE
(tests/golden/speaker_small.npz from tests/golden/make_golden_speaker.py): state_dict contract on the CPU, logits / teacher-forced loss /
gradients / greedy decoding on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
VOCAB, FEAT = 300, 768 + 128


def _gold():
    return np.load(os.path.join(HERE, 'golden', 'speaker_small.npz'))


def _model():
    from vln_goat_amd import speaker
    cfg = speaker.default_config(speaker_dropout=0.0, featdropout=0.0)
    torch.manual_seed(0)
    return speaker.Transpeaker(FEAT, 512, 256, VOCAB, cfg)


def test_speaker_state_dict_matches_the_reference():
    z = _gold()
    m = _model()
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in z['param_names']]
    assert [n for n, _ in m.named_parameters()] == [str(k) for k in z['nomask_grad_names']]
    assert sd['encoder.pos_emb.pe'].shape == (5000, 1, 512) and sd['decoder.pos_emb.pe'].shape == (5000, 1, 256)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_speaker_matches_reference_golden(dtype):
    import make_golden_speaker as mg
    import vln_goat_amd
    from vln_goat_amd import speaker
    z = _gold()
    m = _model()
    m.load_state_dict(mg.seeded_state(m.state_dict()))
    m = m.cuda().eval()
    can, img, insts, ctx_mask = mg.inputs()
    can, img, insts_t = torch.from_numpy(can).cuda(), torch.from_numpy(img).cuda(), torch.from_numpy(insts).cuda()
    tol = 1e-3 if dtype == torch.float32 else 2e-2
    vln_goat_amd.set_compute_dtype(dtype)
    try:
        for tag, cm in (('nomask', None), ('ctxmask', torch.from_numpy(ctx_mask).cuda())):
            for p in m.parameters():
                p.grad = None
            logits = m(can, img, insts_t, ctx_mask=cm, already_dropfeat=True)
            ref = z[tag + '_logits']
            assert np.abs(logits.detach().float().cpu().numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max()), tag
            loss = speaker.teacher_forcing_loss(m, can, img, insts_t, pad_id=0, ctx_mask=cm)
            assert abs(float(loss) - float(z[tag + '_loss'][0])) <= tol * float(z[tag + '_loss'][0]), tag
            loss.backward()
            fp = z[tag + '_grad_fp']
            top = float(fp[:, 0].max())
            params = dict(m.named_parameters())
            for n, refp in zip([str(k) for k in z[tag + '_grad_names']], fp):
                g = params[n].grad
                norm = 0.0 if g is None else float(g.double().norm())
                rt = 2e-3 if dtype == torch.float32 else 6e-2
                assert abs(norm - float(refp[0])) <= rt * max(float(refp[0]), 1e-3 * top), (tag, n, norm, float(refp[0]))
        if dtype == torch.float32:
            words = speaker.infer_batch(m, can, img, bos=1, eos=2, pad=0, unk=3, max_decode=10, already_dropfeat=True)
            assert np.array_equal(words.cpu().numpy(), z['greedy_words'])
    finally:
        vln_goat_amd.set_compute_dtype(torch.float32)


@pytest.mark.gpu
def test_speaker_path_features_and_decoding_on_the_navigator():
    """path_features walks the ground-truth paths on the graph-only navigator; shapes, zero rows at the stop step, and a decode."""
    import vln_goat_amd
    from vln_goat_amd import features, rollout, speaker, synth
    scan, feats, eps, _ = synth.make_rollout_case()
    store = features.FeatureStore.from_arrays({'%s_%s' % (scan.name, vp): feats[i] for i, vp in enumerate(scan.vpids)}, dtype=torch.bfloat16).to('cuda')
    sim = rollout.GraphSim(store)
    img, can, lengths = speaker.path_features(sim, store, eps)
    B, T = len(eps), max(len(e['path']) for e in eps)
    assert img.shape == (B, T, 36, FEAT) and can.shape == (B, T, FEAT) and lengths.tolist() == [len(e['path']) for e in eps]
    for b, e in enumerate(eps):
        stop = len(e['path']) - 1
        assert not bool(can[b, stop].any()) and bool(can[b, 0].any())
        row = scan.index[e['path'][0]]
        assert torch.equal(img[b, 0, :, :768].cpu(), torch.from_numpy(feats[row]).to(torch.bfloat16).float())
    m = _model().cuda().eval()
    vln_goat_amd.set_compute_dtype(torch.bfloat16)
    try:
        words = speaker.infer_batch(m, can, img, bos=1, eos=2, pad=0, unk=3, max_decode=6)
        assert words.shape[0] == B and 2 <= words.shape[1] <= 7 and int(words[:, 0].min()) == 1 and not bool((words == 3).any())
    finally:
        vln_goat_amd.set_compute_dtype(torch.float32)
