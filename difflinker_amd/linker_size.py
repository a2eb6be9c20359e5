"""Linker-size predictor — drop-ins for ``src/linker_size.py`` (``SizeGNN``, ``DistributionNodes``) and for the
inference side of ``src/linker_size_lightning.py`` (``SizeClassifier``), the ``sample_fn`` that ``generate.py:86-99``
runs once before a sampling chain.

The modules hold the parameters under the reference's ``state_dict`` keys; the arithmetic runs in the HIP kernel of
``csrc/size_gnn.hip`` through the C ABI (``dl_size_gnn_forward``).  There is no PyTorch fallback: CPU tensors raise.
Training (gradients, BatchNorm in train mode, the ordinal / regression variants) is out of scope.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib, const
from .egnn import GCL

try:  # pragma: no cover
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:
    _Base = nn.Module


class DistributionNodes:
    """Categorical over linker sizes from a histogram (linker_size.py:9-42)."""

    def __init__(self, histogram):
        self.n_nodes = []
        prob = []
        self.keys = {}
        for i, nodes in enumerate(histogram):
            self.n_nodes.append(nodes)
            self.keys[nodes] = i
            prob.append(histogram[nodes])
        self.n_nodes = torch.tensor(self.n_nodes)
        prob = np.array(prob)
        prob = prob / np.sum(prob)
        self.prob = torch.from_numpy(prob).float()
        self.m = torch.distributions.Categorical(torch.tensor(prob))

    def sample(self, n_samples=1):
        return self.n_nodes[self.m.sample((n_samples,))]

    def log_prob(self, batch_n_nodes):
        assert len(batch_n_nodes.size()) == 1
        idcs = torch.tensor([self.keys[i.item()] for i in batch_n_nodes]).to(batch_n_nodes.device)
        return torch.log(self.prob + 1e-30).to(batch_n_nodes.device)[idcs]


class _HipSizeModel:
    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            if self.handle:
                _lib.load().dl_size_model_destroy(self.handle)
        except Exception:  # pragma: no cover - interpreter shutdown
            pass


class SizeGNN(nn.Module):
    """``SizeGNN(in_node_nf, hidden_nf, out_node_nf, n_layers, normalization, device)`` — linker_size.py:45-81."""

    def __init__(self, in_node_nf, hidden_nf, out_node_nf, n_layers, normalization, device='cpu'):
        super().__init__()
        if hidden_nf != 128:
            raise NotImplementedError('the HIP path is built for hidden_nf = 128 (the reference default)')
        if normalization not in (None, 'batch_norm'):
            raise NotImplementedError(normalization)
        self.hidden_nf = hidden_nf
        self.out_node_nf = out_node_nf
        self.in_node_nf = in_node_nf
        self.n_layers = n_layers
        self.normalization = normalization
        self.device = device

        def make_gcl():
            return GCL(input_nf=hidden_nf, output_nf=hidden_nf, hidden_nf=hidden_nf, normalization_factor=1,
                       aggregation_method='sum', edges_in_d=1, activation=nn.ReLU(), attention=False,
                       normalization=normalization)

        self.embedding_in = nn.Linear(in_node_nf, hidden_nf)
        self.gcl1 = make_gcl()
        self.gcl_layers = nn.ModuleList([make_gcl() for _ in range(n_layers - 1)])
        self.embedding_out = nn.Linear(hidden_nf, out_node_nf)
        self._hip = {}
        self.to(device)

    # ------------------------------------------------------------------------------------------------
    def _folded_linear(self, lin, bn):
        """(W, b) of ``bn(lin(x))`` with the BatchNorm in eval mode (running statistics)."""
        w, b = lin.weight.detach().double().cpu(), lin.bias.detach().double().cpu()
        if bn is not None:
            if bn.training:
                raise NotImplementedError('BatchNorm in training mode: call .eval() (training is out of scope)')
            s = bn.weight.detach().double().cpu() / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
            w = w * s[:, None]
            b = (b - bn.running_mean.detach().double().cpu()) * s + bn.bias.detach().double().cpu()
        return w.float().contiguous(), b.float().contiguous()

    def _host_tensors(self):
        ts = [self.embedding_in.weight.detach().float().cpu().contiguous(),
              self.embedding_in.bias.detach().float().cpu().contiguous()]
        for gcl in [self.gcl1] + list(self.gcl_layers):
            ts += [gcl.edge_mlp[0].weight, gcl.edge_mlp[0].bias, gcl.edge_mlp[2].weight, gcl.edge_mlp[2].bias]
            if self.normalization is None:
                ts += list(self._folded_linear(gcl.node_mlp[0], None)) + list(self._folded_linear(gcl.node_mlp[2], None))
            else:
                ts += list(self._folded_linear(gcl.node_mlp[0], gcl.node_mlp[1]))
                ts += list(self._folded_linear(gcl.node_mlp[3], gcl.node_mlp[4]))
        ts += [self.embedding_out.weight, self.embedding_out.bias]
        return [t.detach().float().cpu().contiguous() for t in ts]

    def _version(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def hip_model(self, device):
        index = device.index if device.index is not None else torch.cuda.current_device()
        version = self._version()
        cached = self._hip.get(index)
        if cached is not None and cached[1] == version:
            return cached[0].handle
        lib = _lib.load()
        cfg = _lib.DLSizeConfig(self.in_node_nf, self.hidden_nf, self.out_node_nf, self.n_layers)
        ts = self._host_tensors()
        n = lib.dl_size_model_num_tensors(ctypes.byref(cfg))
        _lib.check(min(n, 0), 'dl_size_model_num_tensors')
        assert n == len(ts), (n, len(ts))
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        handle = ctypes.c_void_p()
        with torch.cuda.device(index):
            _lib.check(lib.dl_size_model_create(ctypes.byref(cfg), ptrs, n, ctypes.byref(handle)), 'dl_size_model_create')
        self._hip[index] = (_HipSizeModel(handle), version)
        return handle

    def _launch(self, one_hot, positions, fragment_mask, edge_mask, distances):
        dev = one_hot.device
        if dev.type != 'cuda':
            raise _lib.HipLibraryError('SizeGNN runs on the HIP device only (no CPU fallback): move the inputs to cuda')
        B, N = one_hot.shape[:2]
        f32 = lambda t: None if t is None else t.to(torch.float32).contiguous()   # noqa: E731
        one_hot, positions, distances = f32(one_hot), f32(positions), f32(distances)
        fragment_mask = f32(fragment_mask.reshape(B, N))
        edge_mask = f32(edge_mask.reshape(B, N, N))
        logits = torch.empty((B, self.out_node_nf), dtype=torch.float32, device=dev)
        flags = torch.zeros((B,), dtype=torch.int32, device=dev)
        handle = self.hip_model(dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.load().dl_size_gnn_forward(handle, B, N, _lib.ptr(one_hot), _lib.ptr(positions),
                                                       _lib.ptr(fragment_mask), _lib.ptr(edge_mask), _lib.ptr(distances),
                                                       _lib.ptr(logits), _lib.ptr(flags), stream),
                       'dl_size_gnn_forward')
        fl = flags.cpu()
        if bool((fl & 4).any()):
            raise ValueError(f'more than {_lib.load().dl_size_max_fragment_atoms()} fragment atoms in a molecule: '
                             'outside the LDS-resident size-predictor kernel')
        return logits

    def predict_logits(self, one_hot, positions, fragment_mask, edge_mask):
        """Fused inference entry: masks, squared distances, the ``< 6`` edge filter, the GNN and the node mean in one
        launch (what ``SizeClassifier.forward`` computes, linker_size_lightning.py:83-110)."""
        return self._launch(one_hot, positions, fragment_mask, edge_mask, None)

    def forward(self, h, edges, distances, node_mask, edge_mask):
        """Reference signature (linker_size.py:83-91): flattened ``h [B*N, in]``, the fully-connected edge list
        ``[rows, cols]`` (e = b*N*N + i*N + j), per-edge ``distances [E,1]`` and the final ``edge_mask [E,1]``.
        The reference returns per-node outputs ``[B*N, out]``; the HIP kernel only materialises their per-molecule
        mean (all any caller uses), so this low-level entry is not provided."""
        raise NotImplementedError('per-node outputs are not materialised by the HIP kernel; call '
                                  'SizeClassifier.forward(data, return_loss=False) or SizeGNN.predict_logits')


class SizeClassifier(_Base):
    """Inference drop-in of ``SizeClassifier`` (linker_size_lightning.py:14-117)."""

    def __init__(self, data_path=None, train_data_prefix=None, val_data_prefix=None, in_node_nf=None, hidden_nf=128,
                 out_node_nf=None, n_layers=3, batch_size=64, lr=1e-3, torch_device='cpu', normalization=None,
                 loss_weights=None, min_linker_size=None, linker_size2id=const.ZINC_TRAIN_LINKER_SIZE2ID,
                 linker_id2size=const.ZINC_TRAIN_LINKER_ID2SIZE, task='classification'):
        super().__init__()
        if hasattr(self, 'save_hyperparameters') and _Base is not nn.Module:  # pragma: no cover
            self.save_hyperparameters()
        self.data_path = data_path
        self.train_data_prefix = train_data_prefix
        self.val_data_prefix = val_data_prefix
        self.min_linker_size = min_linker_size
        self.linker_size2id = linker_size2id
        self.linker_id2size = linker_id2size
        self.batch_size = batch_size
        self.lr = lr
        self.loss_weights = loss_weights
        self.in_node_nf = in_node_nf
        self.gnn = SizeGNN(in_node_nf=in_node_nf, hidden_nf=hidden_nf, out_node_nf=out_node_nf, n_layers=n_layers,
                           device='cpu', normalization=normalization)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **overrides):
        if _Base is not nn.Module:  # pragma: no cover
            return super().load_from_checkpoint(checkpoint_path, map_location=map_location, strict=strict, **overrides)
        ckpt = torch.load(checkpoint_path, map_location=map_location or 'cpu', weights_only=False)
        hparams = dict(ckpt['hyper_parameters'])
        hparams.update(overrides)
        model = cls(**hparams)
        model.load_state_dict(ckpt['state_dict'], strict=strict)
        return model

    def forward(self, data, return_loss=True, with_pocket=False, adjust_shape=False):
        """``(logits [B, out_node_nf], loss)`` — linker_size_lightning.py:83-117."""
        h = data['one_hot']
        x = data['positions']
        fragment_mask = data['fragment_only_mask'] if with_pocket else data['fragment_mask']
        edge_mask = data['edge_mask']
        if h.shape[-1] != self.in_node_nf and adjust_shape:
            assert torch.allclose(h[..., -1] * fragment_mask[..., 0], torch.zeros_like(h[..., -1]))
            h = h[..., :-1]
        bs, n_nodes = x.shape[0], x.shape[1]
        if 'edges' in data:
            rows = data['edges'][0]
            assert rows.numel() == bs * n_nodes * n_nodes, 'the HIP path expects the fully-connected edge list of collate_with_fragment_edges'
        output = self.gnn.predict_logits(h, x, fragment_mask, edge_mask)
        loss = None
        if return_loss:
            # sample.py:71 calls forward() with the default return_loss=True and discards the loss; it is a [B, classes]
            # cross-entropy on the HIP logits (no gradient: training is out of scope)
            weight = None if self.loss_weights is None else torch.as_tensor(self.loss_weights, device=output.device)
            loss = torch.nn.functional.cross_entropy(output, self.get_true_labels(data['linker_mask']), weight=weight)
        return output, loss

    def get_true_labels(self, linker_mask):
        """Class index of every molecule's true linker size; unseen sizes map to the largest class
        (linker_size_lightning.py:119-129)."""
        labels = []
        for size in linker_mask.reshape(linker_mask.shape[0], -1).sum(-1).long().detach().cpu().numpy():
            label = self.linker_size2id.get(int(size))
            if label is None:
                label = self.linker_size2id[max(self.linker_id2size)]
            labels.append(label)
        return torch.tensor(labels, device=linker_mask.device, dtype=torch.long)

    def sample_sizes(self, data, with_pocket=False):
        """The ``sample_fn`` closure of generate.py:86-99 as a method: softmax -> Categorical -> size table."""
        out, _ = self.forward(data, return_loss=False, with_pocket=with_pocket)
        probabilities = torch.softmax(out, dim=1)
        samples = torch.distributions.Categorical(probs=probabilities).sample()
        sizes = [self.linker_id2size[label] for label in samples.detach().cpu().numpy()]
        return torch.tensor(sizes, device=samples.device, dtype=const.TORCH_INT)
