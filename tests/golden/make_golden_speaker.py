"""Golden vectors for the speaker FROM THE IMPORTED REFERENCE (this container only): `Transpeaker`
(/root/reference/map_nav_src/models/transpeaker_model.py:232-257) with seeded weights on seeded inputs, eval mode, speaker_dropout 0
(the reference's attention dropout lives in a module built inside forward and cannot be switched off otherwise): logits, the
teacher-forced loss of M/r2r/transpeaker.py:233-239 and the gradient fingerprint of every parameter.  The module parses sys.argv at
import (M/r2r/parser.py) and calls .cuda() in constructors: argv is replaced, .cuda() patched to the identity.
    python tests/golden/make_golden_speaker.py    -> tests/golden/speaker_small.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

VOCAB, B, T, L = 300, 3, 4, 12
FEAT = 768 + 128


def inputs():
    rs = np.random.RandomState(23)
    can = rs.standard_normal((B, T, FEAT)).astype(np.float32)
    img = rs.standard_normal((B, T, 36, FEAT)).astype(np.float32)
    insts = np.zeros((B, L), np.int64)
    for b, n in enumerate((12, 9, 6)):
        insts[b, 0] = 1                                  # <BOS>
        insts[b, 1:n - 1] = rs.randint(4, VOCAB, n - 2)
        insts[b, n - 1] = 2                              # <EOS>; the rest stays <PAD> = 0
    ctx_mask = np.zeros((B, T), bool)
    ctx_mask[1, 3:] = True
    ctx_mask[2, 2:] = True
    return can, img, insts, ctx_mask


def seeded_state(sd, seed=31):
    rs = np.random.RandomState(seed)
    out = {}
    for k, v in sd.items():
        if k.endswith('pos_emb.pe'):
            out[k] = v.clone()
        else:
            out[k] = torch.from_numpy((rs.standard_normal(tuple(v.shape)) * 0.05).astype(np.float32))
    return out


def fingerprint(g):
    flat = g.detach().float().reshape(-1)
    first = torch.zeros(8)
    first[:min(8, flat.numel())] = flat[:8]
    return np.concatenate([[float(flat.double().norm())], first.numpy()]).astype(np.float32)


def main():
    ref_shim._install_common()
    sys.argv = ['speaker', '--speaker_dropout', '0.0', '--featdropout', '0.0', '--mode', 'train']
    p = ref_shim.REF_ROOT + '/map_nav_src'
    sys.path.insert(0, p)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    import models.transpeaker_model as tm
    model = tm.Transpeaker(feature_size=FEAT, hidden_size=512, word_size=256, tgt_vocab_size=VOCAB)
    sd = seeded_state(model.state_dict())
    model.load_state_dict(sd)
    model.eval()
    can, img, insts, ctx_mask = inputs()
    store = {'param_names': np.array(list(sd.keys()))}
    for tag, cm in (('nomask', None), ('ctxmask', torch.from_numpy(ctx_mask))):
        for p_ in model.parameters():
            p_.grad = None
        logits, _, _, _ = model(torch.from_numpy(can.copy()), torch.from_numpy(img.copy()), torch.from_numpy(insts), ctx_mask=cm, already_dropfeat=True)
        loss = torch.nn.functional.cross_entropy(logits.permute(0, 2, 1)[:, :, :-1], torch.from_numpy(insts)[:, 1:], ignore_index=0)
        loss.backward()
        store[tag + '_logits'] = logits.detach().numpy()
        store[tag + '_loss'] = np.array([float(loss)], np.float32)
        store[tag + '_grad_fp'] = np.stack([fingerprint(p_.grad) for _, p_ in model.named_parameters()])
        store[tag + '_grad_names'] = np.array([n for n, _ in model.named_parameters()])
    # greedy decoding (M/r2r/transpeaker.py:270-312) from the same encoder states
    with torch.no_grad():
        enc_inputs, enc_outputs, _ = model.encoder(torch.from_numpy(can.copy()), torch.from_numpy(img.copy()), already_dropfeat=True)
        word = torch.ones(B, 1, dtype=torch.int64)
        ended = np.zeros(B, bool)
        for _ in range(10):
            dec, _, _ = model.decoder(word, enc_inputs, enc_outputs)
            lg = model.projection(dec)
            lg[:, :, 3] = -float('inf')                  # <UNK>
            nxt = lg.max(dim=-1)[1][:, -1]
            nxt[torch.from_numpy(ended)] = 0
            word = torch.cat([word, nxt.unsqueeze(-1)], -1)
            ended = np.logical_or(ended, nxt.numpy() == 2)
            if ended.all():
                break
    store['greedy_words'] = word.numpy()
    path = os.path.join(HERE, 'speaker_small.npz')
    np.savez_compressed(path, **store)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB; losses', float(store['nomask_loss'][0]), float(store['ctxmask_loss'][0]), 'greedy', word.shape)


if __name__ == '__main__':
    main()
