"""Dataset sampling drivers — the bodies of the reference's ``sample.py`` (:26-200) and ``sample_trajectories.py``
(:19-110) as importable functions: walk a preprocessed validation / test set, write the ground truth, the
fragments (and the pocket) once, then ``n_samples`` sampled molecules per entry as ``<uuid>/<i>_.xyz``; resumable the
way the script is (``check_if_generated``).  The matplotlib rendering of ``sample_trajectories.py`` (``visualize_chain``)
is presentation and is not provided; the per-frame ``.xyz`` files it renders from are.

``python -m difflinker_amd.sample --help`` exposes ``sample.py``'s flags; ``--keep_frames K`` switches to the
trajectory mode.
"""
import argparse
import os

import torch

from . import utils
from .datasets import MOADDataset, collate, collate_with_fragment_edges, get_dataloader
from .io import save_xyz_file
from .lightning import DDPM
from .linker_size import SizeClassifier


def check_if_generated(_output_dir, _uuids, n_samples):
    """``(everything already there, first sample index to (re)generate)`` — sample.py:37-60, including its
    restart-two-back rule."""
    generated = True
    starting_points = []
    for _uuid in _uuids:
        numbers = []
        for fname in os.listdir(os.path.join(_output_dir, _uuid)):
            try:
                numbers.append(int(fname.split('_')[0]))
            except ValueError:
                continue
        if len(numbers) == 0 or max(numbers) != n_samples - 1:
            generated = False
            starting_points.append(0 if len(numbers) == 0 else max(numbers) - 1)
    starting = min(starting_points) if len(starting_points) > 0 else None
    return generated, starting


def _prepare(checkpoint, prefix, data, n_steps, device):
    model = checkpoint if isinstance(checkpoint, DDPM) else DDPM.load_from_checkpoint(checkpoint, map_location=device)
    model.val_data_prefix = prefix
    if data is not None:
        model.data_path = data
    if n_steps is not None:
        model.edm.T = n_steps
    model = model.eval().to(device)
    model.torch_device = device
    model.setup(stage='val')
    return model


def sample(checkpoint, samples, prefix, n_samples, device, data=None, n_steps=None, linker_size_model=None):
    """``sample.py``.  Returns the output directory."""
    exp = 'model' if isinstance(checkpoint, DDPM) else checkpoint.split('/')[-1].replace('.ckpt', '')
    collate_fn, sample_fn = collate, None
    if linker_size_model is None:
        output_dir = os.path.join(samples, prefix, exp)
    else:
        if isinstance(linker_size_model, SizeClassifier):
            size_nn, size_name = linker_size_model, 'size_model'
        else:
            size_nn = SizeClassifier.load_from_checkpoint(linker_size_model, map_location=device)
            size_name = linker_size_model.split('/')[-1].replace('.ckpt', '')
        size_nn = size_nn.eval().to(device)
        output_dir = os.path.join(samples, prefix, 'sampled_size', size_name, exp)
        collate_fn = collate_with_fragment_edges

        def sample_fn(_data):                                  # sample.py:70-80 (long sizes, loss discarded)
            output, _ = size_nn.forward(_data)
            samples_ = torch.distributions.Categorical(probs=torch.softmax(output, dim=1)).sample()
            sizes = [size_nn.linker_id2size[label] for label in samples_.detach().cpu().numpy()]
            return torch.tensor(sizes, device=samples_.device, dtype=torch.long)
    os.makedirs(output_dir, exist_ok=True)

    model = _prepare(checkpoint, prefix, data, n_steps, device)
    moad = isinstance(model.val_dataset, MOADDataset)
    for batch_idx, batch in enumerate(model.val_dataloader(collate_fn=collate_fn)):
        uuids = [str(u) for u in batch['uuid']]
        for u in uuids:
            os.makedirs(os.path.join(output_dir, u), exist_ok=True)
        generated, starting_point = check_if_generated(output_dir, uuids, n_samples)
        if generated:
            continue
        h, x, node_mask, frag_mask = batch['one_hot'], batch['positions'], batch['atom_mask'], batch['fragment_mask']
        if moad and model.center_of_mass == 'fragments':
            com_mask = batch['fragment_only_mask']
        elif model.center_of_mass == 'fragments':
            com_mask = batch['fragment_mask']
        elif model.center_of_mass == 'anchors':
            com_mask = batch['anchors']
        else:
            raise NotImplementedError(model.center_of_mass)
        x = utils.remove_partial_mean_with_mask(x, node_mask, com_mask)
        if moad:
            node_mask = batch['atom_mask'] - batch['pocket_mask']
            frag_mask = batch['fragment_only_mask']
            save_xyz_file(output_dir, h, x, batch['pocket_mask'], [f'{u}/pock' for u in uuids], is_geom=model.is_geom)
        save_xyz_file(output_dir, h, x, node_mask, [f'{u}/true' for u in uuids], is_geom=model.is_geom)
        save_xyz_file(output_dir, h, x, frag_mask, [f'{u}/frag' for u in uuids], is_geom=model.is_geom)
        for i in range(starting_point, n_samples):
            chain, out_mask = model.sample_chain(batch, sample_fn=sample_fn, keep_frames=1)
            xs, hs = chain[0][:, :, :model.n_dims], chain[0][:, :, model.n_dims:]
            if moad:
                pock = batch['pocket_mask']
                if pock.shape[1] < out_mask.shape[1]:          # template wider than the input (sampled sizes)
                    pock = torch.nn.functional.pad(pock, (0, 0, 0, out_mask.shape[1] - pock.shape[1]))
                out_mask = out_mask - pock
            save_xyz_file(output_dir, hs, xs, out_mask, [f'{u}/{i}' for u in uuids], is_geom=model.is_geom)
    return output_dir


def sample_trajectories(checkpoint, chains, prefix, keep_frames, device, data=None, n_steps=None, batch_size=32):
    """``sample_trajectories.py``: ``keep_frames`` frames of every chain as ``chains/<k>/<k>_<frame>_.xyz`` plus the
    final prediction and the ground truth under ``final_states/``.  The feature slice ``3:-1`` is the script's own
    (it assumes a trailing charge column)."""
    exp = 'model' if isinstance(checkpoint, DDPM) else checkpoint.split('/')[-1].replace('.ckpt', '')
    chains_dir = os.path.join(chains, exp, prefix, 'chains')
    final_dir = os.path.join(chains, exp, prefix, 'final_states')
    os.makedirs(chains_dir, exist_ok=True)
    os.makedirs(final_dir, exist_ok=True)
    model = _prepare(checkpoint, prefix, data, n_steps, device)
    start = 0
    for batch in get_dataloader(model.val_dataset, batch_size=batch_size):
        chain_batch, node_mask = model.sample_chain(batch, keep_frames=keep_frames)
        for i in range(len(batch['positions'])):
            chain = chain_batch[:, i, :, :]
            assert chain.shape[0] == keep_frames and chain.shape[1] == batch['positions'].shape[1]
            name = str(i + start)
            out = os.path.join(chains_dir, name)
            os.makedirs(out, exist_ok=True)
            frames_mask = torch.cat([node_mask[i].unsqueeze(0) for _ in range(keep_frames)], dim=0)
            save_xyz_file(out, chain[:, :, 3:-1], chain[:, :, :3], frames_mask,
                          names=[f'{name}_{j}' for j in range(keep_frames)], is_geom=model.is_geom)
            save_xyz_file(final_dir, batch['one_hot'][i].unsqueeze(0), batch['positions'][i].unsqueeze(0),
                          batch['atom_mask'][i].unsqueeze(0), names=[f'{name}_true'], is_geom=model.is_geom)
            save_xyz_file(final_dir, chain[0, :, 3:-1].unsqueeze(0), chain[0, :, :3].unsqueeze(0),
                          frames_mask[0].unsqueeze(0), names=[f'{name}_pred'], is_geom=model.is_geom)
        start += len(batch['positions'])
    return chains_dir, final_dir


def main(argv=None):
    p = argparse.ArgumentParser(description='DiffLinker dataset sampling on MI355X (sample.py / sample_trajectories.py)')
    p.add_argument('--checkpoint', required=True)
    p.add_argument('--samples', required=True, help='output root (sample.py --samples / sample_trajectories.py --chains)')
    p.add_argument('--data', default=None)
    p.add_argument('--prefix', required=True)
    p.add_argument('--n_samples', type=int, default=1)
    p.add_argument('--n_steps', type=int, default=None)
    p.add_argument('--linker_size_model', default=None)
    p.add_argument('--keep_frames', type=int, default=None, help='trajectory mode: frames kept per chain')
    p.add_argument('--device', default='cuda:0')
    a = p.parse_args(argv)
    if a.keep_frames is not None:
        print(sample_trajectories(a.checkpoint, a.samples, a.prefix, a.keep_frames, a.device, a.data, a.n_steps))
    else:
        print(sample(a.checkpoint, a.samples, a.prefix, a.n_samples, a.device, a.data, a.n_steps, a.linker_size_model))


if __name__ == '__main__':
    main()
