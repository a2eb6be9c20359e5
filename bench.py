"""bench.py — molecules/s of the full 500-step ``sample_chain`` (BASELINE.json metric) on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``; started without a
   launcher, ``python bench.py --gpus N`` starts its N ranks itself the same way)

A "step" is one complete ``EDM.sample_chain`` over one synthetic batch: T = 500 reverse steps + the
final decode = 501 EGNN forwards, per GPU the workload of BASELINE config C2 (GEOM hparams, 6 blocks,
B = 256 molecules padded to N = 50, n_b ~ U{35..50}).  Weak scaling: every rank samples its own 256
molecules (global batch 256*N, config C3 at N = 8); no data-path collective, one all-gather of the
final frame (RCCL over xGMI) inside the timed region.  Inputs are resident in HBM before the timed
region; the noise is drawn inside the kernel (counter-based Philox keyed by the global molecule index, the
same source at every --gpus; --noise torch = the reference's ``torch.randn`` call sequence, a secondary line).

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     — MFMA roofline of the dominant kernel (``sample_chain_fc_kernel``): the FLOPs the kernel
                 executes x 501 forwards / live HIP-event duration of the launch (the reference algorithm's
                 count, SURVEY 8d F_min, is the `algorithmic` side entry)
  cpu_baseline — the oracle (PyTorch-CPU port of the reference path) timed on this box's host cores
                 on a bounded sample (2 forwards of the same batch, extrapolated x501).
  secondary    — driver-timed companions: exact-fp32 mode, torch noise, C4, C5 (one GPU's shard), molecules of
                 60..80 atoms, other batch sizes.
  --config C5 / C4 / C1 / C2L select another workload as the headline; --backend gloo runs --gpus N as N ranks
  on ONE GPU (functional check of the sharded branch on a single-GPU box).
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / tensor sharing fail with hipIpcGetMemHandle otherwise); the
# driver exports it already - kept here so that a hand-started torchrun behaves the same
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector peak
F16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense BF16/F16 MFMA peak (spec)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='C2', choices=['C1', 'C2', 'C4', 'C5', 'C2L', 'C2XL'])
    ap.add_argument('--batch', type=int, default=None, help='molecules per GPU (default: the config\'s)')
    ap.add_argument('--T', type=int, default=None, help='reverse steps (default: the config\'s, 500 for C2)')
    ap.add_argument('--uniform-size', action='store_true', help='unpadded variant: every molecule has N atoms')
    ap.add_argument('--noise', default='philox', choices=['torch', 'philox'],
                    help="'philox' (default, every --gpus): counter-based draws generated inside the kernel, keyed by the global "
                         "molecule index - what a batch sharded over GPUs needs; 'torch': the reference's torch.randn call "
                         "sequence (1004 launches per chain on the host's stream; N = 1 only, reported as a secondary line)")
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend of --gpus > 1: 'nccl' = RCCL over xGMI (default); 'gloo' lets several ranks "
                         "share ONE GPU (functional check of the sharded path on a single-GPU box; RCCL refuses that)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-forwards', type=int, default=2)
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the secondary measurements (exact-fp32 mode, C4 pockets, other batch sizes) of the N=1 line')
    ap.add_argument('--precision', default=None, choices=['f16x3', 'fp32', 'f16x2'], help='arithmetic mode (default: f16x3)')
    ap.add_argument('--traffic-child', action='store_true', help=argparse.SUPPRESS)     # the profiled child of measure_traffic()
    ap.add_argument('--no-traffic', action='store_true', help='skip the two rocprofv3 --pmc passes behind roofline.traffic')
    return ap.parse_args()


def build_model(cfg, device):
    from difflinker_amd import Dynamics, DynamicsWithPockets, EDM
    torch.manual_seed(0)                                   # random-init weights of the named architecture
    cls = Dynamics if cfg['graph_type'] == 'FC' else DynamicsWithPockets
    dyn = cls(n_dims=3, in_node_nf=cfg['nf'], context_node_nf=cfg['ctx'], hidden_nf=int(cfg.get('hidden_nf', 128)),
              n_layers=cfg['n_layers'], norm_constant=1e-6, normalization='batch_norm', graph_type=cfg['graph_type'],
              sin_embedding=bool(cfg.get('sin_embedding', False)))
    edm = EDM(dyn, in_node_nf=cfg['nf'], n_dims=3, timesteps=cfg.get('timesteps', 500), noise_schedule='polynomial_2',
              noise_precision=1e-5, loss_type='l2', norm_values=[1, 4, 10])
    edm.T = cfg['T']
    if cfg.get('precision'):
        dyn.precision = cfg['precision']
    return edm.to(device)


def pocket_edge_count(inp, cutoff_cross=10.0):
    """Directed edges of the FC-10A-4A radius graph on the batch's input coordinates (bookkeeping for the
    algorithmic FLOP count only; egnn.py:565-596)."""
    x = inp['x']
    nm = inp['node_mask'].squeeze(-1).bool()
    lig = (inp['linker_mask'].squeeze(-1).bool() | inp['context'][..., -2].bool()) & nm
    poc = inp['context'][..., -1].bool() & nm
    d = torch.cdist(x, x)
    eye = torch.eye(x.shape[1], dtype=torch.bool).unsqueeze(0)
    adj = (lig[:, :, None] & lig[:, None, :]) | (poc[:, :, None] & poc[:, None, :] & (d <= 4)) | \
          (((lig[:, :, None] & poc[:, None, :]) | (poc[:, :, None] & lig[:, None, :])) & (d <= cutoff_cross))
    adj = adj & nm[:, :, None] & nm[:, None, :] & ~eye
    into_linker = adj & inp['linker_mask'].squeeze(-1).bool()[:, :, None]       # receiving atom i of edge (i, j) is a linker atom
    return int(adj.sum()), int(into_linker.sum())


REFERENCE_ROOT = '/root/reference'     # the unmodified reference: present in the build container, absent on the GPU box


def reference_forward(edm, cfg):
    """``Dynamics.forward`` of the UNMODIFIED reference (src/egnn.py:374-447 / :471-552) with this model's weights, or None
    where the reference tree does not exist (the GPU box: the oracle port - the same numbers to the bit, 4 % apart in time,
    profiles/r04/cpu_reference_vs_port.log - is timed instead)."""
    if not os.path.isfile(os.path.join(REFERENCE_ROOT, 'src', 'egnn.py')):
        return None
    try:
        sys.dont_write_bytecode = True                         # the reference tree is read-only
        sys.path.insert(0, REFERENCE_ROOT)
        from src import egnn as ref_egnn
        pockets = cfg['graph_type'] != 'FC'
        cls = ref_egnn.DynamicsWithPockets if pockets else ref_egnn.Dynamics
        ref = cls(n_dims=3, in_node_nf=cfg['nf'], context_node_nf=cfg['ctx'], hidden_nf=128, device='cpu', n_layers=cfg['n_layers'],
                  attention=False, tanh=False, norm_constant=1e-6, inv_sublayers=2, sin_embedding=False, normalization_factor=100,
                  aggregation_method='sum', model='egnn_dynamics', normalization='batch_norm', centering=False,
                  graph_type=cfg['graph_type'])
        ref.load_state_dict({k: v.detach().cpu().clone() for k, v in edm.dynamics.state_dict().items()}, strict=True)
        ref.eval()
        return lambda sd, ocfg, t, z, nm, lm, em, ctx: ref.forward(t=t, xh=z, node_mask=nm, linker_mask=lm, edge_mask=em, context=ctx)
    except Exception as e:                                     # a reference tree that does not import here: the port
        print(f'note: the reference modules under {REFERENCE_ROOT} could not be used ({type(e).__name__}: {e}); timing the port',
              file=sys.stderr)
        return None
    finally:
        if sys.path and sys.path[0] == REFERENCE_ROOT:
            sys.path.pop(0)


def cpu_baseline(edm, cfg, inp, n_forwards, sample_batch=32):
    """The reference path on the host cores, bounded sample: the first `sample_batch` molecules of the same batch, best of a
    few thread counts (PyTorch CPU ops stop scaling well before a many-core host is full), scaled linearly to the whole
    batch and to T+1 forwards (per-edge cost is constant; every step costs the same).  What is timed: the unmodified
    reference modules where /root/reference exists (``kind: "reference"``), else the oracle = the PyTorch-CPU port of the
    same op sequence (``kind: "port"``)."""
    from oracle import egnn_oracle
    cores = os.cpu_count() or 1
    sd = {k: v.detach().cpu().clone() for k, v in edm.dynamics.state_dict().items()}
    pockets = cfg['graph_type'] != 'FC'
    ocfg = egnn_oracle.EGNNConfig(in_node_nf=cfg['nf'], context_node_nf=cfg['ctx'], n_layers=cfg['n_layers'],
                                  graph_type=cfg['graph_type'])
    forward = reference_forward(edm, cfg)
    kind = 'reference' if forward is not None else 'port'
    if forward is None:
        forward = egnn_oracle.dynamics_forward_pockets if pockets else egnn_oracle.dynamics_forward
    B, N = inp['x'].shape[:2]
    b = min(8 if pockets else sample_batch, B)
    g = torch.Generator().manual_seed(1)
    z = torch.cat([inp['x'], inp['h']], dim=2) * inp['fragment_mask'] + \
        torch.randn((B, N, 3 + cfg['nf']), generator=g) * inp['linker_mask']
    t = torch.full((b, 1), 0.5)
    em = inp['edge_mask'].view(B, N)[:b].reshape(-1) if pockets else inp['edge_mask'].view(B, N * N)[:b].reshape(-1, 1)
    args = (sd, ocfg, t, z[:b], inp['node_mask'][:b], inp['linker_mask'][:b], em, inp['context'][:b])
    best = None
    with torch.no_grad():
        for threads in sorted({min(8, cores), min(16, cores), min(32, cores), min(64, cores)}):
            torch.set_num_threads(threads)
            forward(*args)                                 # warm-up (edge list, allocator)
            t0 = time.perf_counter()
            for _ in range(n_forwards):
                forward(*args)
            dt = (time.perf_counter() - t0) / n_forwards
            if best is None or dt < best[0]:
                best = (dt, threads)
    dt, threads = best
    fwd_full = dt * B / b
    chain_s = fwd_full * (cfg['T'] + 1)
    what = 'Dynamics.forward calls of the unmodified reference (src/egnn.py)' if kind == 'reference' else \
        'Dynamics.forward calls of the oracle port (the reference\'s op sequence on PyTorch-CPU; /root/reference is absent here)'
    return {'value': B / chain_s, 'unit': 'molecules/s', 'cores': threads, 'kind': kind,
            'sample': f'{n_forwards} {what} on the first {b} of the {B} molecules (N={N}, '
                      f'L={cfg["n_layers"]}) after 1 warm-up, best of 8/16/32/64 threads ({threads}): {dt:.2f} s each; '
                      f'scaled x{B / b:g} to the batch and x{cfg["T"] + 1} to the chain; host has {cores} logical cores',
            's_per_forward_full_batch': fwd_full}


def measure_traffic(a, cfg, child_T=50):
    """Fabric-side bytes per launch of the dominant kernel, MEASURED BY THIS RUN (round 4 read them from a
    committed file): two extra ``rocprofv3 --pmc`` passes (FETCH_SIZE, then WRITE_SIZE - they do not fit one pass,
    MI355X_MICROARCH.md) over a child process that samples the same batch with T = `child_T`, after the timed region.  The
    per-forward figure (every forward costs the same) is scaled to the T + 1 forwards of the benchmarked launch; FETCH_SIZE is
    doubled (gfx950 tallies 128-byte requests as 64) and both are KiB.  Returns (bytes per launch or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None:
        return None, 'not measured: rocprofv3 is not on PATH'
    per_forward = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        out_dir = tempfile.mkdtemp(prefix=f'dl_pmc_{counter}_', dir='/tmp')
        cmd = ['rocprofv3', '--pmc', counter, '-d', out_dir, '--output-format', 'csv', '--', sys.executable, os.path.abspath(__file__),
               '--traffic-child', '--config', a.config, '--T', str(child_T), '--noise', a.noise, '--steps', '1', '--warmup', '1',
               '--no-secondary', '--no-cpu-baseline'] + (['--batch', str(a.batch)] if a.batch is not None else []) + \
              (['--precision', a.precision] if a.precision else []) + (['--uniform-size'] if a.uniform_size else [])
        # (counter collection of ROCm 7.2 crashes on cooperative launches - the second launch of a split chain is one: the child
        # issues the same kernel, grid and arguments through the plain launch API)
        env = dict(os.environ, TMPDIR='/tmp', DIFFLINKER_TEAM_LAUNCH_PLAIN='1')
        try:
            subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
            vals = []
            for path in glob.glob(os.path.join(out_dir, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(path)):
                    if 'sample_chain_fc_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter:
                        vals.append(float(r['Counter_Value']))
            if not vals:
                return None, f'not measured: the {counter} pass recorded no launch of sample_chain_fc_kernel'
            # the child samples 2 chains (1 warm-up + 1); a chain may be two launches (EDM.split_chain): per chain, not per launch
            per_forward[counter] = sum(vals) / 2.0 * 1024.0 / (child_T + 1)
        except Exception as e:
            return None, f'not measured: the rocprofv3 --pmc {counter} pass failed ({type(e).__name__})'
        finally:
            shutil.rmtree(out_dir, ignore_errors=True)
    fetch, write = 2.0 * per_forward['FETCH_SIZE'], per_forward['WRITE_SIZE']
    total = (fetch + write) * (cfg['T'] + 1)
    return total, (f'measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes) over a child process sampling the same '
                   f'batch with T = {child_T}; per forward {fetch / 1e6:.0f} MB fetched (FETCH_SIZE x2, gfx950) + {write / 1e6:.0f} MB '
                   f'written, x{cfg["T"] + 1} forwards; fabric-side, Infinity-Cache hits included (the fp32 node-feature tiles a '
                   f'workgroup parks in its HBM scratch and the L2 misses of the weight stream); the algorithmic bytes are ~2 MB per forward')


def time_chains(edm, inp, steps=1, warmup=1):
    """(seconds per chain, kernel ms or None) of ``edm.sample_chain`` on resident inputs."""
    edm.profile_events = True
    for _ in range(warmup):
        edm.sample_chain(keep_frames=1, **inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = []
    for _ in range(steps):
        edm.sample_chain(keep_frames=1, **inp)
        ev = getattr(edm, 'last_kernel_events', None)
        kms.append(ev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    k = None
    if kms and kms[0] is not None:
        k = sum(s.elapsed_time(e) for s, e in kms) / len(kms)
    return dt, k


def secondary_measurements(device, a):
    """Driver-timed companions of the headline: the exact-fp32 arithmetic mode on the same C2 batch, the
    pocket configuration C4, and C2 at batch sizes off the one-molecule-per-compute-unit sweet spot.  One warm-up chain
    and one timed chain each (a chain is 501 forwards: the timing noise is well below 1 %).  Every line carries the fraction of
    its ceiling on the reference algorithm's flop count (``roofline_frac``, SURVEY 8d F_min) AND on the work the kernels execute
    (``executed_frac``: the coordinate head only for receiving atoms inside the linker mask) - VERDICT round 3."""
    from difflinker_amd import synthetic
    out = []
    kept = {}

    def run(tag, config, batch, precision, note, team='auto', noise=None, sin_embedding=False, hidden_nf=128):
        data, cfg = synthetic.make_batch(config, seed=1000, batch=batch)
        cfg['precision'] = precision
        cfg['sin_embedding'] = sin_embedding
        cfg['hidden_nf'] = hidden_nf                      # (the flop counts below are those of the 128-wide kernels that run it)
        pockets = cfg['graph_type'] != 'FC'
        inp_cpu = synthetic.sampler_inputs(data, pockets=pockets)
        inp = {k: v.to(device) for k, v in inp_cpu.items()}
        edm = build_model(cfg, device)
        edm.noise_source = noise or a.noise
        edm.dynamics.team = team
        if hasattr(edm, 'last_kernel_events'):
            del edm.last_kernel_events
        torch.manual_seed(4321)
        dt, kms = time_chains(edm, inp)
        pairs, nodes = synthetic.pair_and_node_counts(data)
        pairs_coord = synthetic.coord_pair_count(data)
        if pockets:
            pairs, pairs_coord = pocket_edge_count(inp_cpu)
        fin = cfg['nf'] + cfg['ctx'] + 1
        flops = synthetic.flops_min(128, cfg['n_layers'], fin, pairs, nodes) * (cfg['T'] + 1)
        flops_exec = synthetic.flops_executed(128, cfg['n_layers'], fin, pairs, pairs_coord, nodes) * (cfg['T'] + 1)
        terms = synthetic.split_terms(precision, 128, cfg['n_layers'], fin, pairs, pairs_coord, nodes)
        peak = FP32_MFMA_PEAK_TFLOPS if precision == 'fp32' else F16_MFMA_PEAK_TFLOPS / terms
        t_k = (kms * 1e-3) if kms is not None else dt
        B = inp['x'].shape[0]
        fused = (not pockets) and kms is not None
        if tag == 'c4_pockets' and not a.no_cpu_baseline:
            # C4's own CPU baseline: DynamicsWithPockets.forward (src/egnn.py:471-552) of the reference
            # where it exists, else of the port, on 8 of the 64 molecules
            try:
                kept['c4_cpu_baseline'] = cpu_baseline(edm, cfg, inp_cpu, 1)
            except Exception as e:
                kept['c4_cpu_baseline'] = {'value': None, 'error': f'{type(e).__name__}: {e}'}
        out.append({'tag': tag, 'workload': f'{config}, batch={B}, T={cfg["T"]}, {precision}, noise={edm.noise_source}; {note}',
                    'compute_units_per_molecule': None if (pockets or not fused) else (max(2, edm.dynamics.team_for_size(B, device)) if config == 'C2L' else edm.dynamics.team_for(B)), 'molecules_per_s': B / dt, 'ms_per_chain': 1e3 * dt, 'kernel_ms': kms,
                    'roofline_frac': flops / t_k / 1e12 / peak, 'executed_frac': flops_exec / t_k / 1e12 / peak,
                    'roofline_peak_tflops': peak, 'achieved_tflops': flops / t_k / 1e12, 'executed_tflops': flops_exec / t_k / 1e12})

    run('c2_f16x2', 'C2', None, 'f16x2', 'the headline batch in the opt-in two-term arithmetic of the GCL edge models (Dynamics.precision = \'f16x2\': '
        'node features 3..9e-6 rel-L2 per forward against the fp32 oracle instead of 2..5e-7, coordinates unchanged; DESIGN.md)')
    run('c2_fp32_mode', 'C2', None, 'fp32', 'exact fp32 MFMA arithmetic (v_mfma_f32_32x32x2_f32), same batch as the headline')
    run('c2_torch_noise', 'C2', None, 'f16x3', "the headline batch with the reference's torch.randn call sequence as the noise source "
        '(2 x 502 randn launches per chain on the stream, a 308 MB bank in HBM) instead of the in-kernel draws', noise='torch')
    run('c4_pockets', 'C4', None, 'f16x3', 'pockets_difflinker_full_no_anchors_fc, N=292, FC-10A-4A radius graph')
    run('c4_pockets_f16x2', 'C4', None, 'f16x2', 'C4 in the opt-in two-term arithmetic')
    run('c5_shard', 'C5', None, 'f16x3', 'BASELINE config 5, one GPU\'s shard: the C4 molecules, batch 64, EDM built with timesteps = 1000, T = 1000')
    run('c2_large_molecules', 'C2L', None, 'f16x3', '60..80 atoms per molecule: beyond one compute unit\'s LDS (55), fused chain on teams of at least two '
        'compute units per molecule (each holds its own atoms\' state and every atom\'s sender row); round 2: HBM-resident kernels, host loop')
    run('c2_xl_molecules_host_loop', 'C2XL', None, 'f16x3', '120..150 atoms per molecule: beyond the fused paths (110), HBM-resident per-pass kernels '
        '(dl_egnn_forward_fc_large) under the host-driven 501-step loop')
    run('c2_sin_embedding_host_loop', 'C2', 64, 'f16x3', 'a sin_embedding model (egnn.py:281-292; no released configuration): every molecule on the '
        'HBM-resident kernels under the host-driven loop', sin_embedding=True)
    run('c2_batch_64_one_cu_each', 'C2', 64, 'f16x3', 'the reference\'s default sampling batch (generate.py:145), one compute unit per '
        'molecule: a quarter of the chip', team=1)
    for b in (64, 128, 257, 320, 512):
        run(f'c2_batch_{b}', 'C2', b, 'f16x3', 'Dynamics.team = auto: 4 / 2 compute units per molecule while the batch leaves the chip '
            'room (atoms dealt round-robin, per-atom phases split, sender rows exchanged through HBM once per pass), else one, biggest first; '
            'round 5: up to 1.25x the number of compute units, the molecules beyond one per compute unit are sampled by teams in a second, '
            'concurrent launch that takes the compute units the smallest molecules leave early (B = 257 was 373..389 molecules/s)')
    run('c2_hidden_64', 'C2', None, 'f16x3', "the reference's DEFAULT width hidden_nf = 64 (egnn.py:324-329; round 5: narrower networks run "
        'zero-padded on the 128-wide kernels - the same function at the 128-wide cost; fractions count the 128-wide work)', hidden_nf=64)
    try:
        kept['size_gnn'] = time_size_gnn(device)
    except Exception as e:
        kept['size_gnn'] = {'ms_per_call': None, 'error': f'{type(e).__name__}: {e}'}
    return out, kept


def time_size_gnn(device, batch=64, n_calls=20):
    """``dl_size_gnn_forward`` (csrc/size_gnn.hip: the linker-size predictor ``sample_fn`` of generate.py:110-128, SURVEY 8 row
    f3) timed once: the reference's default sampling batch, GEOM-sized fragments, 5 layers - it runs once per batch, before
    the chain."""
    from difflinker_amd.datasets import collate_with_fragment_edges
    from difflinker_amd.linker_size import SizeClassifier
    from difflinker_amd import synthetic
    torch.manual_seed(0)
    clf = SizeClassifier(in_node_nf=9, hidden_nf=128, out_node_nf=20, n_layers=5, normalization='batch_norm').eval().to(device)
    mols = synthetic.fc_molecules(batch, 50, 35, (3, 12), 9, seed=1000)
    data = collate_with_fragment_edges(mols)
    data = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}
    with torch.no_grad():
        clf.forward(data, return_loss=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_calls):
            clf.forward(data, return_loss=False)
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n_calls
    return {'ms_per_call': ms, 'workload': f'SizeClassifier.forward (dl_size_gnn_forward), batch {batch}, N = 50, 5 layers, hidden 128; wall time of '
                                           f'the Python call, {n_calls} calls after 1 warm-up'}


def eager_rocm_baseline(edm, cfg, inp_cpu, device, n_forwards=3):
    """The reference path as plain PyTorch on THIS GPU (ROCm eager, fp32): the oracle port (``oracle/egnn_oracle.py``, the
    reference's op sequence: edge list, gather, cat, linear, SiLU, scatter-add) moved to ``cuda:0`` - what a user of the
    reference gets on an MI355X without this library (SURVEY 8d lists it as the optional second baseline).  A few forwards of
    the whole batch after one warm-up, scaled to the T + 1 forwards of a chain; a baseline, never the product path."""
    from oracle import egnn_oracle
    pockets = cfg['graph_type'] != 'FC'
    sd = {k: v.detach().to(device).clone() for k, v in edm.dynamics.state_dict().items()}
    ocfg = egnn_oracle.EGNNConfig(in_node_nf=cfg['nf'], context_node_nf=cfg['ctx'], n_layers=cfg['n_layers'],
                                  graph_type=cfg['graph_type'])
    forward = egnn_oracle.dynamics_forward_pockets if pockets else egnn_oracle.dynamics_forward
    B, N = inp_cpu['x'].shape[:2]
    g = torch.Generator().manual_seed(1)
    z = torch.cat([inp_cpu['x'], inp_cpu['h']], dim=2) * inp_cpu['fragment_mask'] + \
        torch.randn((B, N, 3 + cfg['nf']), generator=g) * inp_cpu['linker_mask']
    t = torch.full((B, 1), 0.5)
    args = [sd, ocfg] + [v.to(device) for v in (t, z, inp_cpu['node_mask'], inp_cpu['linker_mask'], inp_cpu['edge_mask'],
                                                  inp_cpu['context'])]
    with torch.no_grad():
        forward(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_forwards):
            forward(*args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_forwards
    chain_s = dt * (cfg['T'] + 1)
    return {'value': B / chain_s, 'unit': 'molecules/s', 'kind': 'port', 'device': torch.cuda.get_device_name(device),
            'sample': f'{n_forwards} Dynamics.forward calls of the oracle port (plain PyTorch fp32, ROCm eager) on the whole batch of {B} after '
                      f'1 warm-up: {1e3 * dt:.1f} ms each; scaled x{cfg["T"] + 1} to the chain (every step costs the same; the sampler '
                      f'algebra of a step is not counted)', 'ms_per_forward': 1e3 * dt}


def self_spawn(a):
    """``python bench.py --gpus N`` started WITHOUT a launcher (the driver's command shape for N = 1): start the
    N ranks ourselves through ``torch.distributed.run`` (one process per GPU, rendezvous on 127.0.0.1, a free port) and hand
    their output through; rank 0 of the child job prints the one JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // a.gpus)))
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_spawn(a))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == a.gpus or (a.gpus == 1 and world == 1), f'--gpus {a.gpus} but WORLD_SIZE={world}'
    if world > 1 and a.backend == 'gloo':                  # several ranks on one GPU (functional check): the rendezvous needs no device,
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')  # so it - and one collective - runs before the device check (CPU test of --gpus 8)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        seen = torch.ones(1)
        dist.all_reduce(seen)
        print(f'rank {rank}/{world} joined ({int(seen)} ranks in the group)', file=sys.stderr, flush=True)
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback)'
    device = torch.device('cuda', local_rank if a.backend == 'nccl' else local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    if world > 1 and a.backend == 'nccl':
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()                                      # (re)compile once; the other ranks wait for it
    if world > 1:
        dist.barrier()
    if rank != 0:
        entry.build()
    from difflinker_amd import synthetic

    from difflinker_amd.distributed import sample_chain_sharded, shard_bounds
    # weak scaling: the logical batch has 256 molecules per GPU (config C3 at N = 8); every rank builds the same logical
    # batch from one seed and samples its contiguous shard (distributed.sample_chain_sharded), so the N-GPU job computes
    # exactly what one GPU would for that batch (in-kernel counter-based noise keyed by the GLOBAL molecule index)
    per_gpu = a.batch if a.batch is not None else synthetic.CONFIGS[a.config]['batch']
    data, cfg = synthetic.make_batch(a.config, seed=1000, batch=per_gpu * world, uniform_size=a.uniform_size)
    if a.T is not None:
        cfg['T'] = a.T
    cfg['precision'] = a.precision
    pockets = cfg['graph_type'] != 'FC'
    inp_cpu = synthetic.sampler_inputs(data, pockets=pockets)
    inp = {k: v.to(device) for k, v in inp_cpu.items()}    # inputs resident in HBM before the timed region
    Bg, N = inp['x'].shape[:2]
    lo, hi = shard_bounds(Bg, rank, world)
    B = hi - lo
    edm = build_model(cfg, device)
    assert world == 1 or a.noise == 'philox', 'a batch sharded over ranks needs the counter-based noise (--noise philox)'
    edm.noise_source = a.noise
    edm.profile_events = True
    shard_cpu = {k: (v[lo:hi] if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == Bg else v) for k, v in inp_cpu.items()}
    pairs, nodes = synthetic.pair_and_node_counts({'atom_mask': data['atom_mask'][lo:hi]})
    pairs_coord = synthetic.coord_pair_count({'atom_mask': data['atom_mask'][lo:hi], 'linker_mask': data['linker_mask'][lo:hi]})
    if pockets:
        pairs, pairs_coord = pocket_edge_count(shard_cpu)
    fin = cfg['nf'] + cfg['ctx'] + 1
    flops_fwd = synthetic.flops_min(128, cfg['n_layers'], fin, pairs, nodes)
    flops_fwd_exec = synthetic.flops_executed(128, cfg['n_layers'], fin, pairs, pairs_coord, nodes)

    def one_chain():
        if world == 1:
            return edm.sample_chain(keep_frames=1, **inp)
        return sample_chain_sharded(edm, inp, keep_frames=1)      # one all-gather of the final frame (RCCL)

    torch.manual_seed(1234 + rank)
    for _ in range(a.warmup):
        one_chain()
    kernel_ms = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_chain()
        kernel_ms.append(getattr(edm, 'last_kernel_events', None))
        split_events = getattr(edm, 'last_split_event', None)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    if kernel_ms[0] is not None:
        k_ms = [s.elapsed_time(e) for s, e in kernel_ms]
        k_avg_ms = sum(k_ms) / len(k_ms)
    else:                                                  # pocket path: many launches per chain
        k_avg_ms = 1e3 * elapsed / a.steps
    per_rank = None
    if world > 1:                                          # after the timed region: every rank's kernel time and all-gather time
        from difflinker_amd.distributed import last_gather_ms
        mine = {'rank': rank, 'kernel_ms': k_avg_ms, 'all_gather_ms': last_gather_ms(), 'molecules': B}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        achieved = flops_fwd * (cfg['T'] + 1) / (k_avg_ms * 1e-3) / 1e12
        executed = flops_fwd_exec * (cfg['T'] + 1) / (k_avg_ms * 1e-3) / 1e12
        layer_bytes = synthetic.layer_bytes(nodes, pairs)
        t_layer = k_avg_ms * 1e-3 / ((cfg['T'] + 1) * cfg['n_layers'])
        precision = edm.dynamics.precision
        if precision in ('f16x3', 'f16x2'):
            # the 128-wide contractions run as 3 fp16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate): the
            # ALGORITHMIC-flop ceiling of the scheme is the dense f16 MFMA peak / 3 (f16x2: 2 terms in the GCL edge models'
            # second layer, 3 elsewhere - the average over the executed work, synthetic.split_terms)
            terms = synthetic.split_terms(precision, 128, cfg['n_layers'], fin, pairs, pairs_coord, nodes)
            peak, peak_note = F16_MFMA_PEAK_TFLOPS / terms, \
                f'dense f16 MFMA peak 2500 TFLOP/s / {terms:.2f} split terms per multiply-accumulate; the pair loop (71 % of a forward) is bound by the ' \
                'issue port its VALU, transcendental and matrix instructions share (profiles/r04); the per-atom phases (21 %; round 6: an ' \
                'atom-stationary GEMM chain with LDS-streamed weights, profiles/r06) by chains of latencies, see DESIGN.md'
        else:
            peak, peak_note = FP32_MFMA_PEAK_TFLOPS, 'v_mfma_f32_32x32x2_f32 = fp32 vector peak'
        # fabric-side bytes per launch of the dominant kernel: two rocprofv3 --pmc passes of a T = 50 child, outside the timed
        # region (measure_traffic); null with the reason when they could not run
        traffic, traffic_note = None, 'not measured (pocket path: many kernels per chain; multi-rank runs; --no-traffic)'
        if not pockets and world == 1 and not a.no_traffic and not a.traffic_child:
            traffic, traffic_note = measure_traffic(a, cfg)
        DTYPES = {'f16x3': 'f32 (f16x3 split MFMA, fp32 accumulate)', 'fp32': 'f32',
                  'f16x2': 'f32 (f16x2 / f16x3 split MFMA, fp32 accumulate)'}
        out = {
            'metric': 'molecules/sec (500-step sample_chain)', 'value': Bg * a.steps / elapsed,
            'unit': 'molecules/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': 1e3 * elapsed / a.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': DTYPES[precision], 'precision': precision, 'data': 'synthetic', 'noise': edm.noise_source,
            'config': {'workload': f'{a.config}: {"GEOM geom_difflinker" if not pockets else "pockets_difflinker_full_no_anchors_fc (FC-10A-4A radius graph)"} hparams (egnn_dynamics, hidden 128, '
                                   f'{cfg["n_layers"]} blocks), batch={B} molecules/GPU padded to N={N} '
                                   f'(n_b {"= N" if a.uniform_size else ("~ U{35..50}" if not pockets else "30 fragment + 250 pocket + 6..12 linker atoms")}), T={cfg["T"]} reverse steps '
                                   f'+ decode = {cfg["T"] + 1} EGNN forwards per step; random-init weights, synthetic '
                                   f'fragment graphs',
                       'global_batch': Bg, 'molecules_per_gpu': B, 'n_nodes': N, 'T': cfg['T'], 'parallelism': f'batch-shard x{world}' + ('' if world == 1 else ' (distributed.sample_chain_sharded: contiguous shards, in-kernel Philox noise keyed by the global molecule index, one RCCL all-gather of the final frame)'),
                       'real_pairs_per_forward': pairs, 'coordinate_pass_pairs_per_forward': pairs_coord, 'real_atoms': nodes},
            # primary figures: the work the kernels EXECUTE: the coordinate head runs for receiving atoms
            # inside the linker mask only - the reference multiplies every other atom's sum by zero (egnn.py:113-116) - so
            # the reference algorithm's flop count (SURVEY 8d F_min: every pair in all three edge models) overstates how
            # busy the hardware is; it stays as the `algorithmic` side entry
            'roofline': {'bound': 'mfma', 'achieved': executed, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': executed / peak, 'traffic': traffic, 'traffic_note': traffic_note, 'peak_note': peak_note,
                         'frac_of_fp32_vector_peak': executed / FP32_MFMA_PEAK_TFLOPS,
                         # the same executed work over the DENSE f16 MFMA peak, split terms not credited
                         'frac_of_dense_f16_peak': executed / F16_MFMA_PEAK_TFLOPS,
                         'kernel': ('sample_chain_fc_kernel' + (' - two launches per chain (EDM.split_chain: <1,false,false> for every molecule, then '
                                                               '<1,true,false> for the big ones on teams of two; kernel_ms = both, see split_chain)'
                                                               if getattr(edm, 'split_chain', False) and split_events is not None else ''))
                         if not pockets else 'all kernels of the chain (pk_edge_kernel dominates)', 'kernel_ms': k_avg_ms,
                         'flops_per_launch': flops_fwd_exec * (cfg['T'] + 1),
                         'counts': 'executed work: GCL edge models on every pair, coordinate edge model on the ' + str(pairs_coord) + ' of '
                                   + str(pairs) + ' pairs per pass whose receiving atom is inside the linker mask, per-atom GEMMs',
                         'algorithmic': {'achieved': achieved, 'frac': achieved / peak, 'flops_per_launch': flops_fwd * (cfg['T'] + 1),
                                         'note': 'SURVEY 8d F_min x (T + 1): the reference algorithm, every pair in all three edge models '
                                                 'of a block / the same kernel time'},
                         # kept for continuity with rounds 1-2, whose lines carried the executed figures under this key
                         'executed': {'achieved': executed, 'frac': executed / peak, 'flops_per_launch': flops_fwd_exec * (cfg['T'] + 1)}},
            'split_chain': None if not getattr(edm, 'split_chain', False) or split_events is None or kernel_ms[0] is None else
                {'first_launch_ms': kernel_ms[-1][0].elapsed_time(split_events), 'second_phase_ms': split_events.elapsed_time(kernel_ms[-1][1])},
            'per_rank': per_rank,
            'per_rank_note': None if per_rank is None else
                'all_gather_ms is measured on each rank from the moment ITS chain is enqueued-complete to the end of the collective: it '
                'contains the wait for the slowest rank (a rank whose kernel ends early shows the difference of the kernel times here - '
                'round 4: 99 ms on the rank that finished first against 5.5 ms on the last, two gloo ranks sharing one GPU, whose kernels '
                'run one after the other) plus the transfer itself (<= 1 MB per rank)',
            'hbm_layer': {'achieved': layer_bytes / t_layer / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                          'frac': layer_bytes / t_layer / 1e9 / HBM_PEAK_GBS,
                          'note': 'EGNN-layer algorithmic bytes (SURVEY 8d A_layer) / time per block; the molecule is '
                                  'LDS-resident for the whole chain, so the HBM fraction is << 1 % by design'},
        }
        if a.traffic_child:                                 # the profiled child of measure_traffic(): the launches are all it is for
            return
        # the optional companions must not cost the line: an error in one of them is recorded, not raised
        c4_cpu = None
        if world == 1 and not a.no_secondary and a.config == 'C2' and not a.uniform_size and a.batch is None and a.T is None:
            out['secondary'], kept = secondary_measurements(device, a)
            # the companions the driver's record must not lose (its `parsed` keeps the contract keys,
            # `roofline` and `cpu_baseline` and drops everything else): small top-level keys AND a copy inside `roofline`
            by_tag = {x['tag']: x for x in out['secondary']}
            comp = {}
            for key, tag, field in (('fp32_mode_molecules_per_s', 'c2_fp32_mode', 'molecules_per_s'),
                                    ('fp32_mode_frac_of_fp32_mfma_peak', 'c2_fp32_mode', 'executed_frac'),
                                    ('f16x2_molecules_per_s', 'c2_f16x2', 'molecules_per_s'),
                                    ('c4_molecules_per_s', 'c4_pockets', 'molecules_per_s'),
                                    ('c4_executed_frac', 'c4_pockets', 'executed_frac'),
                                    ('c5_shard_molecules_per_s', 'c5_shard', 'molecules_per_s'),
                                    ('c2_batch_64_molecules_per_s', 'c2_batch_64', 'molecules_per_s'),
                                    ('c2_batch_64_executed_frac', 'c2_batch_64', 'executed_frac'),
                                    ('c2_batch_128_molecules_per_s', 'c2_batch_128', 'molecules_per_s'),
                                    ('c2_batch_257_molecules_per_s', 'c2_batch_257', 'molecules_per_s'),
                                    ('c2_batch_512_molecules_per_s', 'c2_batch_512', 'molecules_per_s'),
                                    ('c2_large_molecules_per_s', 'c2_large_molecules', 'molecules_per_s')):
                if tag in by_tag:
                    comp[key] = by_tag[tag][field]
            comp['size_gnn_ms_per_call'] = kept.get('size_gnn', {}).get('ms_per_call')
            out.update(comp)
            out['roofline']['companions'] = comp
            out['size_gnn'] = kept.get('size_gnn')
            c4_cpu = kept.get('c4_cpu_baseline')
            # the library's default noise source is the reference's torch.randn call sequence (EDM.noise_source = 'torch'): its
            # figure sits at the top level beside `value`, which is measured with the in-kernel draws at every --gpus
            tn = [x for x in out['secondary'] if x['tag'] == 'c2_torch_noise']
            if tn:
                out['value_by_noise_source'] = {'philox (in-kernel, this line)': out['value'],
                                                'torch (reference randn stream, library default)': tn[0]['molecules_per_s']}
        if world == 1 and not a.no_cpu_baseline:
            for key, fn in (('cpu_baseline', lambda: cpu_baseline(edm, cfg, inp_cpu, a.cpu_forwards)),
                            ('eager_rocm_baseline', lambda: eager_rocm_baseline(edm, cfg, inp_cpu, device))):
                try:
                    out[key] = fn()
                except Exception as e:
                    out[key] = {'value': None, 'error': f'{type(e).__name__}: {e}'}
            if c4_cpu is not None and isinstance(out.get('cpu_baseline'), dict):
                out['cpu_baseline']['c4_pockets'] = {k: c4_cpu.get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample', 'error') if k in c4_cpu}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
