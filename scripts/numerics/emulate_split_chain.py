"""The T-step sampling chain under an emulated arithmetic scheme of the GCL passes' second edge layer (see
emulate_split.py): final linker coordinates against the fp32 oracle's chain on the same noise bank - the quantity
tests/test_gpu_parity_hard.py::test_chain_T500_with_a_live_coordinate_head_geom_sized bounds.
Run:  python scripts/numerics/emulate_split_chain.py --scheme f8cross_rn [--coord f16x3] [--T 500] [--gain 0.02]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import edm_oracle, egnn_oracle  # noqa: E402
from emulate_split import Scheme  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scheme', default='f16x3')
    ap.add_argument('--coord', default='f16x3')
    ap.add_argument('--T', type=int, default=500)
    ap.add_argument('--gain', type=float, default=0.02)
    ap.add_argument('--threads', type=int, default=4)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from difflinker_amd import Dynamics, synthetic
    from difflinker_amd.datasets import collate
    import test_gpu_parity as P
    from test_gpu_parity_hard import ragged_fc_molecules
    nf, L, T = 9, 6, args.T
    sizes, linkers = [50, 44, 41, 47], [8, 6, 5, 9]
    sd = P.seeded_state_dict(nf + 1 + 1, 128, L, 96, coord_gain=args.gain)      # the test's weights, without its GPU module
    cfg = egnn_oracle.EGNNConfig(in_node_nf=nf, context_node_nf=1, n_layers=L)
    inp = synthetic.sampler_inputs(collate(ragged_fc_molecules(sizes, linkers, nf, seed=93)))
    B, N = inp['x'].shape[:2]
    bank = edm_oracle.NoiseBank.generate(T, B, N, 3, nf, seed=94)

    def chain():
        orc = edm_oracle.EDMOracle(edm_oracle.make_dynamics_oracle(sd, cfg), in_node_nf=nf, timesteps=500)
        orc.T = T
        bank.reset()
        return orc.sample_chain(inp['x'], inp['h'], inp['node_mask'], inp['fragment_mask'], inp['linker_mask'],
                                inp['edge_mask'], inp['context'], bank, keep_frames=1)
    want = chain()
    orig = egnn_oracle._lin
    sch, coord = Scheme(args.scheme), Scheme(args.coord)

    def lin(p, key, x):
        if key.endswith('edge_mlp.2'):
            return sch.linear(x, p[key + '.weight'], p[key + '.bias'])
        if key.endswith('coord_mlp.2'):
            return coord.linear(x, p[key + '.weight'], p[key + '.bias'])
        return orig(p, key, x)
    egnn_oracle._lin = lin
    got = chain()
    egnn_oracle._lin = orig
    lm = inp['linker_mask']
    ex = float(((got[0][..., :3] - want[0][..., :3]) * lm).norm() / (want[0][..., :3] * lm).norm())
    mism = int((got[0][..., 3:] != want[0][..., 3:]).sum())
    moved = float(((want[0][..., :3] - inp['x']) * lm).norm(dim=-1).max())
    print(f'T={T} gain={args.gain} GCL scheme {args.scheme:10s} coordinate scheme {args.coord:8s}: final linker coordinates '
          f'rel-L2 {ex:.2e} vs the fp32 oracle, one-hot mismatches {mism}, largest displacement {moved:.1f} A', flush=True)


if __name__ == '__main__':
    main()
