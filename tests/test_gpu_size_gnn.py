"""GPU parity of the linker-size predictor (csrc/size_gnn.hip through the C ABI) against the reference fixture and
the CPU oracle (oracle/size_oracle.py).  Tolerance: plain fp32 arithmetic on both sides, different summation order:
rel-L2 <= 1e-5 on the logits."""
import os

import numpy as np
import pytest
import torch

from helpers import rel_l2, max_abs, seeded_size_state_dict
from oracle import size_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dev():
    return torch.device('cuda:0')


def make_classifier(in_nf, out_nf, n_layers, seed, bn):
    from difflinker_amd.linker_size import SizeClassifier
    clf = SizeClassifier(in_node_nf=in_nf, hidden_nf=128, out_node_nf=out_nf, n_layers=n_layers,
                         normalization='batch_norm' if bn else None).eval()
    sd = seeded_size_state_dict(in_nf, 128, out_nf, n_layers, seed=seed, batch_norm=bn, prefix='gnn.')
    clf.load_state_dict(sd, strict=True)
    return clf.to(dev()), sd


def to_dev(d):
    return {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in d.items()}


@pytest.mark.parametrize('tag,n_layers,bn', [('plain', 3, False), ('bn', 2, True)])
def test_size_gnn_matches_reference_fixture(golden_dir, tag, n_layers, bn):
    z = np.load(os.path.join(golden_dir, 'size_gnn.npz'))
    g = {k: torch.from_numpy(z[k]) for k in z.files if z[k].shape != ()}
    clf, _ = make_classifier(8, 10, n_layers, 500 + n_layers, bn)
    data = to_dev({'one_hot': g['one_hot'], 'positions': g['positions'], 'fragment_mask': g['fragment_mask'],
                   'linker_mask': g['linker_mask'], 'edge_mask': g['edge_mask']})
    out, loss = clf.forward(data, return_loss=False)
    assert loss is None and out.shape == (4, 10)
    err = rel_l2(out.cpu(), g['logits_' + tag])
    print(f'[size_gnn {tag}] rel-L2 {err:.3e} max-abs {max_abs(out.cpu(), g["logits_" + tag]):.3e}')
    assert err <= TOL


def random_batch(sizes, linkers, in_nf, seed, scale=1.6):
    from difflinker_amd.datasets import collate_with_fragment_edges
    g = torch.Generator().manual_seed(seed)
    mols = []
    for n, nl in zip(sizes, linkers):
        frag = torch.zeros(n)
        frag[:n - nl] = 1
        types = torch.randint(0, in_nf, (n,), generator=g)
        mols.append({'positions': scale * torch.randn((n, 3), generator=g),
                     'one_hot': torch.nn.functional.one_hot(types, in_nf).float(), 'anchors': torch.zeros(n),
                     'fragment_mask': frag, 'linker_mask': 1 - frag, 'num_atoms': n, 'uuid': 0, 'name': 'm'})
    return collate_with_fragment_edges(mols)


@pytest.mark.parametrize('sizes,linkers', [([1], [0]), ([5, 64, 33, 17], [0, 0, 4, 2]), ([70, 40], [10, 40])])
def test_size_gnn_matches_oracle(sizes, linkers):
    """1 atom; the 64-fragment-atom maximum; a molecule whose atoms are ALL linker (no fragment: logits = bias mean)."""
    in_nf, out_nf, L = 9, 33, 3
    clf, sd = make_classifier(in_nf, out_nf, L, seed=11, bn=False)
    data = random_batch(sizes, linkers, in_nf, seed=sum(sizes))
    ref = size_oracle.size_classifier_logits(sd, data['one_hot'], data['positions'], data['fragment_mask'],
                                             data['edge_mask'], L)
    out, _ = clf.forward(to_dev(data), return_loss=False)
    err = rel_l2(out.cpu(), ref)
    print(f'[size_gnn sizes={sizes}] rel-L2 {err:.3e}')
    assert err <= TOL
    out2, _ = clf.forward(to_dev(data), return_loss=False)
    assert torch.equal(out, out2), 'run-to-run determinism'


def test_size_gnn_distance_filter_and_self_loops_are_observable():
    """Edges are kept where the SQUARED distance < 6 (not the distance) and the diagonal (mask value -2) is a kept
    self loop: the HIP output must follow the oracle through both conventions, and differ when they are dropped."""
    in_nf, out_nf, L = 8, 10, 1
    clf, sd = make_classifier(in_nf, out_nf, L, seed=5, bn=False)
    data = random_batch([12], [0], in_nf, seed=3, scale=1.2)
    out, _ = clf.forward(to_dev(data), return_loss=False)
    ref = size_oracle.size_classifier_logits(sd, data['one_hot'], data['positions'], data['fragment_mask'],
                                             data['edge_mask'], L)
    assert rel_l2(out.cpu(), ref) <= TOL
    no_self = dict(data)
    em = data['edge_mask'].view(1, 12, 12).clone()
    em[0].fill_diagonal_(0)
    no_self['edge_mask'] = em.view(-1, 1)
    out_ns, _ = clf.forward(to_dev(no_self), return_loss=False)
    ref_ns = size_oracle.size_classifier_logits(sd, data['one_hot'], data['positions'], data['fragment_mask'],
                                                no_self['edge_mask'], L)
    assert rel_l2(out_ns.cpu(), ref_ns) <= TOL
    assert rel_l2(out_ns.cpu(), ref) > 1e-3


def test_size_gnn_too_many_fragment_atoms_raises():
    clf, _ = make_classifier(8, 10, 1, seed=1, bn=False)
    data = random_batch([65], [0], 8, seed=1)
    with pytest.raises(ValueError, match='fragment atoms'):
        clf.forward(to_dev(data), return_loss=False)


def test_size_classifier_checkpoint_and_sample_fn(tmp_path):
    """Lightning-style checkpoint round trip and the generate.py:86-99 sample_fn."""
    from difflinker_amd.linker_size import SizeClassifier
    from difflinker_amd import const
    clf, sd = make_classifier(8, 10, 3, seed=21, bn=True)
    path = os.path.join(tmp_path, 'size.ckpt')
    hparams = dict(data_path='d', train_data_prefix='t', val_data_prefix='v', in_node_nf=8, hidden_nf=128, out_node_nf=10,
                   n_layers=3, batch_size=64, lr=1e-3, torch_device='cpu', normalization='batch_norm')
    torch.save({'hyper_parameters': hparams, 'state_dict': {k: v.cpu() for k, v in clf.state_dict().items()}}, path)
    clf2 = SizeClassifier.load_from_checkpoint(path, map_location=dev()).eval().to(dev())
    data = to_dev(random_batch([20, 31], [0, 0], 8, seed=9))
    a, _ = clf.forward(data, return_loss=False)
    b, _ = clf2.forward(data, return_loss=False)
    assert torch.equal(a, b)
    torch.manual_seed(0)
    sizes = clf2.sample_sizes(data)
    assert sizes.dtype == const.TORCH_INT and sizes.shape == (2,)
    assert all(int(s) in const.ZINC_TRAIN_LINKER_ID2SIZE for s in sizes)
