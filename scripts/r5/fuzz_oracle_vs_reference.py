"""Build container only (it imports /root/reference): the oracle against the UNMODIFIED reference modules on random cases - far
more of them than the committed golden fixtures hold.  Random hyper-parameters (widths, depths, GCLs per block, attention, tanh,
mean aggregation, sinusoidal embedding, time feature on / off, norm_constant, normalization_factor, feature and context widths),
fully-connected and radius-graph denoisers, centred denoiser, ragged batches; ``Dynamics.forward`` and the samplers
(``EDM.sample_chain``, ``InpaintingEDM.sample_chain``) with the reference's noise calls replaced by a shared bank, other
schedules and normalisations.  The weights come from tests/helpers.seeded_state_dict and are loaded into the reference modules
with ``load_state_dict(strict=True)``.
    PYTHONDONTWRITEBYTECODE=1 python scripts/r5/fuzz_oracle_vs_reference.py [--cases 300] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True

from src import utils as ref_utils                      # noqa: E402
from src.egnn import Dynamics, DynamicsWithPockets      # noqa: E402
from src.edm import EDM, InpaintingEDM                  # noqa: E402

import test_gpu_parity as P                             # noqa: E402  (input builders only: nothing here touches a GPU)
from helpers import seeded_state_dict, max_abs          # noqa: E402
from oracle import edm_oracle, egnn_oracle              # noqa: E402
from oracle.egnn_oracle import EGNNConfig               # noqa: E402


def draw(seed):
    rng = np.random.default_rng(seed)
    c = dict(seed=seed)
    c['kind'] = str(rng.choice(['forward', 'forward', 'pocket forward', 'chain', 'inpainting chain', 'pocket chain']))
    pocket = c['kind'].startswith('pocket')
    c['nf'] = int(rng.choice([4, 8, 9, 10]))
    c['ctx'] = int(rng.choice([2, 3])) if pocket else int(rng.choice([0, 1, 2]))
    if c['kind'] != 'forward' and c['ctx'] == 0:
        c['ctx'] = 1
    c['L'], c['sub'] = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    c['hidden'] = int(rng.choice([16, 32, 64, 128]))
    c['attention'], c['tanh'] = bool(rng.random() < 0.3), bool(rng.random() < 0.3)
    c['aggregation_method'] = 'mean' if rng.random() < 0.3 else 'sum'
    c['sin_embedding'] = bool(rng.random() < 0.2)
    c['condition_time'] = bool(rng.random() < 0.8)
    c['norm_constant'] = float(rng.choice([0.0, 1e-6, 1.0]))
    c['normalization_factor'] = float(rng.choice([1.0, 10.0, 100.0]))
    c['graph_type'] = str(rng.choice(['4A', 'FC-4A', 'FC-10A-4A'])) if pocket else 'FC'
    c['T'], c['keep'] = int(rng.integers(1, 8)), 1
    c['keep'] = int(rng.integers(1, c['T'] + 1))
    c['schedule'] = str(rng.choice(['polynomial_2', 'polynomial_3', 'polynomial_1']))
    c['precision'] = float(rng.choice([1e-5, 1e-4]))
    c['timesteps'] = int(rng.choice([100, 500, 1000]))
    c['norm_values'] = [float(rng.choice([1.0, 2.0])), float(rng.choice([4.0, 3.0])), 10.0]
    c['norm_bias'] = float(rng.choice([0.0, 0.0, 0.5]))
    nmol = int(rng.integers(1, 5))
    c['sizes'] = [int(rng.integers(2, 30)) for _ in range(nmol)]
    c['linkers'] = [int(rng.integers(1, min(s - 1, 8) + 1)) for s in c['sizes']]
    c['pocket_shape'] = (int(rng.integers(1, 3)), int(rng.integers(4, 12)), int(rng.integers(8, 40)), int(rng.integers(2, 4)))
    return c


def compare(reference, oracle):
    """-> (max-abs difference, max |reference|), or ('both raise', index sets equal?) when the reference raises FoundNaNException"""
    want = got = None
    try:
        want = reference()
    except ref_utils.FoundNaNException as e:
        ref_sets = (e.x_h_nan_idx, e.only_x_nan_idx, e.only_h_nan_idx)
    try:
        got = oracle()
    except egnn_oracle.OracleNaN as e:
        orc_sets = (e.x_h_nan_idx, e.only_x_nan_idx, e.only_h_nan_idx)
    if want is None or got is None:
        if want is None and got is None:
            return 'both raise', ref_sets == orc_sets
        return 'one raises', 'the reference' if want is None else 'the oracle'
    return max_abs(got, want), float(want.abs().max())


def run(c):
    pocket, inpaint = c['kind'].startswith('pocket'), c['kind'].startswith('inpainting')
    flags = dict(attention=c['attention'], tanh=c['tanh'], aggregation_method=c['aggregation_method'], sin_embedding=c['sin_embedding'])
    fin = c['nf'] + c['ctx'] + int(c['condition_time'])
    sd = seeded_state_dict(fin, c['hidden'], c['L'], c['seed'], inv_sublayers=c['sub'], attention=c['attention'],
                           edge_feat_nf=24 if c['sin_embedding'] else 2)
    cls = DynamicsWithPockets if pocket else Dynamics
    dyn = cls(n_dims=3, in_node_nf=c['nf'], context_node_nf=c['ctx'], hidden_nf=c['hidden'], device='cpu', n_layers=c['L'],
              condition_time=c['condition_time'], norm_constant=c['norm_constant'], inv_sublayers=c['sub'],
              normalization_factor=c['normalization_factor'], model='egnn_dynamics', normalization='batch_norm', centering=inpaint,
              graph_type=c['graph_type'], **flags)
    dyn.load_state_dict(sd, strict=True)
    dyn.eval()
    cfg = EGNNConfig(in_node_nf=c['nf'], context_node_nf=c['ctx'], hidden_nf=c['hidden'], n_layers=c['L'], inv_sublayers=c['sub'],
                     condition_time=c['condition_time'], norm_constant=c['norm_constant'], normalization_factor=c['normalization_factor'],
                     graph_type=c['graph_type'], centering=inpaint, **flags)
    g = torch.Generator().manual_seed(c['seed'] + 5)
    if pocket:
        b, nfrag, npock, lo = c['pocket_shape']
        inp, z, t = P.pocket_inputs(batch=b, n_frag=nfrag, n_pocket=npock, linker=(lo, lo + 3), nf=c['nf'], seed=c['seed'] + 2)
    else:
        inp, z, t = P.ragged_inputs(c['sizes'], c['linkers'], c['nf'], seed=c['seed'] + 2)
    B, N = z.shape[:2]
    if c['ctx'] != inp['context'].shape[-1]:
        inp['context'] = (torch.randn(B, N, c['ctx'], generator=g) * inp['node_mask'].float()) if c['ctx'] else None
    if 'forward' in c['kind']:
        fwd = egnn_oracle.dynamics_forward_pockets if pocket else egnn_oracle.dynamics_forward

        def reference():
            with torch.no_grad():
                return dyn.forward(t=t, xh=z, node_mask=inp['node_mask'], linker_mask=inp['linker_mask'], edge_mask=inp['edge_mask'], context=inp['context'])
        return compare(reference, lambda: fwd(sd, cfg, t, z, inp['node_mask'], inp['linker_mask'], inp['edge_mask'], inp['context']))
    T = c['T']
    kw = dict(in_node_nf=c['nf'], n_dims=3, timesteps=c['timesteps'], noise_schedule=c['schedule'], noise_precision=c['precision'],
              norm_values=c['norm_values'], norm_biases=(None, c['norm_bias'], 0.))
    edm = (InpaintingEDM if inpaint else EDM)(dynamics=dyn, loss_type='l2', **kw)
    edm.T = T
    bank = edm_oracle.NoiseBank.generate(2 * T + 1 if inpaint else T, B, N, 3, c['nf'], seed=c['seed'] + 3)
    pos = [0]

    def banked(size, device, node_mask):
        d = bank.draws[pos[0]]
        assert tuple(d.shape) == tuple(size)
        pos[0] += 1
        return d * node_mask
    def banked_randn(size, device=None, **kw):          # InpaintingEDM draws through torch.randn itself (edm.py:715-727)
        d = bank.draws[pos[0]]
        assert tuple(d.shape) == tuple(size)
        pos[0] += 1
        return d.clone()
    orig, real_randn = ref_utils.sample_gaussian_with_mask, torch.randn
    nm = inp['node_mask'].float()
    if inpaint:
        inp['x'] = ref_utils.remove_mean_with_mask(inp['x'] * nm, nm)          # lightning.py:441-446: inpainting centres on all atoms
        torch.randn = banked_randn
    else:
        ref_utils.sample_gaussian_with_mask = banked
    def reference():
        try:
            with torch.no_grad():
                if inpaint:
                    out = edm.sample_chain(x=inp['x'], h=inp['h'], node_mask=nm, edge_mask=inp['edge_mask'], fragment_mask=inp['fragment_mask'],
                                           linker_mask=inp['linker_mask'], context=inp['context'], keep_frames=c['keep'])
                else:
                    out = edm.sample_chain(x=inp['x'], h=inp['h'], node_mask=inp['node_mask'], fragment_mask=inp['fragment_mask'],
                                           linker_mask=inp['linker_mask'], edge_mask=inp['edge_mask'], context=inp['context'], keep_frames=c['keep'])
            assert pos[0] == len(bank.draws)
            return out
        finally:
            ref_utils.sample_gaussian_with_mask, torch.randn = orig, real_randn
    okw = {k: v for k, v in kw.items() if k != 'n_dims'}
    orc = (edm_oracle.InpaintingEDMOracle if inpaint else edm_oracle.EDMOracle)(edm_oracle.make_dynamics_oracle(sd, cfg), **okw)
    orc.T = T

    def oracle():
        bank.reset()
        if inpaint:
            return orc.sample_chain(inp['x'], inp['h'], inp['node_mask'].float(), inp['edge_mask'], inp['fragment_mask'], inp['linker_mask'],
                                    inp['context'], bank, keep_frames=c['keep'])
        return orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'], inp['edge_mask'],
                                inp['context'], bank, keep_frames=c['keep'])
    return compare(reference, oracle)


@torch.no_grad()
def size_predictor(n, bad):
    """oracle/size_oracle.py against the unmodified ``SizeGNN`` (src/linker_size.py:45-91) driven as ``SizeClassifier.forward`` drives it
    at inference (src/linker_size_lightning.py:83-110; tests/golden/make_golden.py::size_gnn restates those ten lines)."""
    from src.linker_size import SizeGNN
    from src.egnn import coord2diff
    from difflinker_amd.datasets import collate_with_fragment_edges
    from helpers import seeded_size_state_dict
    from oracle import size_oracle
    rng = np.random.default_rng(99)
    worst = 0.0
    for k in range(n):
        in_nf, out_nf, L, bn = int(rng.choice([8, 9, 10])), int(rng.integers(2, 40)), int(rng.integers(1, 5)), bool(rng.random() < 0.4)
        g = torch.Generator().manual_seed(1000 + k)
        mols = []
        for _ in range(int(rng.integers(1, 6))):
            frag_n, link = int(rng.integers(0, 40)), int(rng.integers(1, 8))
            nn_ = frag_n + link
            frag = torch.zeros(nn_)
            frag[:frag_n] = 1
            types = torch.randint(0, in_nf, (nn_,), generator=g)
            mols.append({'positions': float(rng.choice([0.5, 1.2, 1.6, 3.0])) * torch.randn((nn_, 3), generator=g),
                         'one_hot': torch.nn.functional.one_hot(types, in_nf).float(), 'anchors': torch.zeros(nn_), 'fragment_mask': frag,
                         'linker_mask': 1 - frag, 'num_atoms': nn_, 'uuid': 0, 'name': 'm'})
        data = collate_with_fragment_edges(mols)
        sd = seeded_size_state_dict(in_nf, 128, out_nf, L, seed=2000 + k, batch_norm=bn)
        gnn = SizeGNN(in_node_nf=in_nf, hidden_nf=128, out_node_nf=out_nf, n_layers=L, normalization='batch_norm' if bn else None)
        gnn.load_state_dict(sd, strict=True)
        gnn.eval()
        fragment_mask, edge_mask, edges = data['fragment_mask'], data['edge_mask'], data['edges']
        x, h = data['positions'] * fragment_mask, data['one_hot'] * fragment_mask
        bs, nn_ = x.shape[0], x.shape[1]
        distances, _ = coord2diff(x.view(bs * nn_, -1), edges)
        dmask = (edge_mask.bool() & (distances < 6)).long()
        want = gnn.forward(h.view(bs * nn_, -1), edges, distances, fragment_mask.view(bs * nn_, 1), dmask).view(bs, nn_, -1).mean(1)
        got = size_oracle.size_classifier_logits(sd, data['one_hot'], data['positions'], data['fragment_mask'], data['edge_mask'], L,
                                                 batch_norm=bn, pre='')
        e = max_abs(got, want)
        worst = max(worst, e)
        if e > 1e-6 * max(1.0, float(want.abs().max())):
            bad.append(f'size predictor case {k}: max-abs {e:.2e}')
    return worst


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=300)
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    t0, worst, bad, kinds, nans = time.time(), {}, [], {}, 0
    if a.seed == 0:
        w = size_predictor(200, bad)
        print(f'linker-size predictor: 200 random cases (8..10 atom types, 2..39 classes, 1..4 layers, with / without BatchNorm), largest difference from the '
              f"reference's SizeGNN {w:.1e}, {len(bad)} failures", flush=True)
    for k in range(a.cases):
        c = draw(a.seed * 100000 + k)
        tag = ' '.join(f'{key}={c[key]}' for key in ('kind', 'nf', 'ctx', 'L', 'sub', 'hidden', 'attention', 'tanh', 'aggregation_method', 'sin_embedding',
                                                       'condition_time', 'norm_constant', 'normalization_factor', 'graph_type'))
        err, mag = run(c)
        if err == 'both raise':
            nans += 1
            print(('ok  ' if mag else 'FAIL'), f'both raise FoundNaNException, index sets equal: {mag}; seed {c["seed"]}: {tag}', flush=True)
            if not mag:
                bad.append(f'index sets differ: seed {c["seed"]}: {tag}')
            continue
        if err == 'one raises':
            bad.append(f'only {mag} raises: seed {c["seed"]}: {tag}')
            print('FAIL', bad[-1], flush=True)
            continue
        rel = err / max(mag, 1e-30)
        ok = err <= 1e-5 * max(1.0, mag) and np.isfinite(err)
        kinds[c['kind']] = kinds.get(c['kind'], 0) + 1
        worst[c['kind']] = max(worst.get(c['kind'], 0.0), rel)
        print(('ok  ' if ok else 'FAIL'), f'max-abs {err:.2e} (|ref| max {mag:.2e})', f'seed {c["seed"]}: {tag}', flush=True)
        if not ok:
            bad.append(f'max-abs {err:.3e} of {mag:.3e}: seed {c["seed"]}: {tag}')
    print(f'{a.cases} cases in {time.time() - t0:.0f} s, {len(bad)} failures; {nans} where both raise FoundNaNException with the same index sets; by kind: ' +
          ', '.join(f'{k}: {n} (worst max-abs / max|ref| {worst[k]:.1e})' for k, n in sorted(kinds.items())))
    for b in bad:
        print('FAILED:', b)
