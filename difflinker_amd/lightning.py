"""``DDPM`` — the wrapper the reference's CLI scripts talk to (``src/lightning.py::DDPM``).

Sampling boundary only: hyper-parameter wiring (lightning.py:39-112), ``load_from_checkpoint`` for
Lightning-format checkpoints (``{'hyper_parameters', 'state_dict'}``), and ``sample_chain``
(:405-463).  Training / validation / metrics (RDKit, WandB, PL Trainer hooks) are out of scope.
Subclasses ``pytorch_lightning.LightningModule`` when that package is importable (it is not in the
build image), else ``torch.nn.Module`` with the same surface the callers use
(generate.py:101-175, sample.py:84-164).
"""
import torch
import torch.nn as nn

from . import utils
from .datasets import MOADDataset, create_templates_for_linker_generation
from .edm import EDM, InpaintingEDM
from .egnn import Dynamics, DynamicsWithPockets

try:  # pragma: no cover
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:
    _Base = nn.Module


def get_activation(activation):
    """lightning.py:27-31."""
    if activation == 'silu':
        return nn.SiLU()
    raise Exception('activation fn not supported yet. Add it here.')


class DDPM(_Base):
    train_dataset = None
    val_dataset = None
    test_dataset = None
    starting_epoch = None
    FRAMES = 100

    def __init__(
        self,
        in_node_nf, n_dims, context_node_nf, hidden_nf, activation, tanh, n_layers, attention, norm_constant,
        inv_sublayers, sin_embedding, normalization_factor, aggregation_method,
        diffusion_steps, diffusion_noise_schedule, diffusion_noise_precision, diffusion_loss_type,
        normalize_factors, include_charges, model,
        data_path, train_data_prefix, val_data_prefix, batch_size, lr, torch_device, test_epochs, n_stability_samples,
        normalization=None, log_iterations=None, samples_dir=None, data_augmentation=False,
        center_of_mass='fragments', inpainting=False, anchors_context=True, graph_type=None,
    ):
        super().__init__()
        self.hparams_dict = {k: v for k, v in locals().items() if k not in ('self', '__class__')}
        if hasattr(self, 'save_hyperparameters') and _Base is not nn.Module:  # pragma: no cover
            self.save_hyperparameters()
        self.data_path = data_path
        self.train_data_prefix = train_data_prefix
        self.val_data_prefix = val_data_prefix
        self.batch_size = batch_size
        self.lr = lr
        self.torch_device = torch_device
        self.include_charges = include_charges
        self.test_epochs = test_epochs
        self.n_stability_samples = n_stability_samples
        self.log_iterations = log_iterations
        self.samples_dir = samples_dir
        self.data_augmentation = data_augmentation
        self.center_of_mass = center_of_mass
        self.inpainting = inpainting
        self.loss_type = diffusion_loss_type
        self.n_dims = n_dims
        self.num_classes = in_node_nf - include_charges
        self.anchors_context = anchors_context
        self.is_geom = ('geom' in self.train_data_prefix) or ('MOAD' in self.train_data_prefix)
        self.pockets = '.' in train_data_prefix              # MOAD prefixes look like 'MOAD_train.full'

        if graph_type is None:
            graph_type = '4A' if self.pockets else 'FC'
        if type(activation) is str:
            activation = get_activation(activation)
        dynamics_class = DynamicsWithPockets if self.pockets else Dynamics
        dynamics = dynamics_class(
            in_node_nf=in_node_nf, n_dims=n_dims, context_node_nf=context_node_nf, device=torch_device,
            hidden_nf=hidden_nf, activation=activation, n_layers=n_layers, attention=attention, tanh=tanh,
            norm_constant=norm_constant, inv_sublayers=inv_sublayers, sin_embedding=sin_embedding,
            normalization_factor=normalization_factor, aggregation_method=aggregation_method, model=model,
            normalization=normalization, centering=inpainting, graph_type=graph_type,
        )
        edm_class = InpaintingEDM if inpainting else EDM           # lightning.py:102
        self.edm = edm_class(
            dynamics=dynamics, in_node_nf=in_node_nf, n_dims=n_dims, timesteps=diffusion_steps,
            noise_schedule=diffusion_noise_schedule, noise_precision=diffusion_noise_precision,
            loss_type=diffusion_loss_type, norm_values=normalize_factors,
        )
        from .const import LINKER_SIZE_DIST
        from .linker_size import DistributionNodes
        self.linker_size_sampler = DistributionNodes(LINKER_SIZE_DIST)          # lightning.py:113

    # ---- checkpoints ----------------------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **overrides):
        """Read a Lightning checkpoint written by the reference's ``ModelCheckpoint``
        (train_difflinker.py:96-101): ``hyper_parameters`` -> constructor, ``state_dict`` ->
        ``load_state_dict(strict)``."""
        if _Base is not nn.Module:  # pragma: no cover
            return super().load_from_checkpoint(checkpoint_path, map_location=map_location, strict=strict, **overrides)
        ckpt = torch.load(checkpoint_path, map_location=map_location or 'cpu', weights_only=False)
        hparams = dict(ckpt['hyper_parameters'])
        hparams.update(overrides)
        model = cls(**hparams)
        model.load_state_dict(ckpt['state_dict'], strict=strict)
        return model

    def checkpoint_dict(self):
        """The ``{'hyper_parameters', 'state_dict'}`` pair ``load_from_checkpoint`` reads."""
        return {'hyper_parameters': dict(self.hparams_dict), 'state_dict': self.state_dict()}

    def setup(self, stage=None):
        """Load the preprocessed dataset(s) (lightning.py:115-137); ``stage='val'`` is what the sampling scripts use."""
        from .datasets import MOADDataset, ZincDataset
        dataset_type = MOADDataset if '.' in self.train_data_prefix else ZincDataset
        if stage == 'fit':
            self.is_geom = ('geom' in self.train_data_prefix) or ('MOAD' in self.train_data_prefix)
            self.train_dataset = dataset_type(data_path=self.data_path, prefix=self.train_data_prefix, device=self.torch_device)
            self.val_dataset = dataset_type(data_path=self.data_path, prefix=self.val_data_prefix, device=self.torch_device)
        elif stage == 'val':
            self.is_geom = ('geom' in self.val_data_prefix) or ('MOAD' in self.val_data_prefix)
            self.val_dataset = dataset_type(data_path=self.data_path, prefix=self.val_data_prefix, device=self.torch_device)
        else:
            raise NotImplementedError

    def train_dataloader(self, collate_fn=None):
        from .datasets import collate, get_dataloader
        return get_dataloader(self.train_dataset, self.batch_size, collate_fn=collate_fn or collate, shuffle=True)

    def val_dataloader(self, collate_fn=None):
        from .datasets import collate, get_dataloader
        return get_dataloader(self.val_dataset, self.batch_size, collate_fn=collate_fn or collate)

    def test_dataloader(self, collate_fn=None):
        from .datasets import collate, get_dataloader
        return get_dataloader(self.test_dataset, self.batch_size, collate_fn=collate_fn or collate)

    def forward(self, data, training):
        raise NotImplementedError('DDPM.forward is the training step (lightning.py:148-199): out of scope')

    # ---- sampling -------------------------------------------------------------------------------------
    def sample_chain(self, data, sample_fn=None, keep_frames=None):
        """``DDPM.sample_chain`` (lightning.py:405-463): linker sizes -> zero templates -> context ->
        fragment-COM removal -> ``EDM.sample_chain``.  Returns ``(chain, node_mask)``."""
        if sample_fn is None:
            linker_sizes = data['linker_mask'].sum(1).view(-1).int()
        else:
            linker_sizes = sample_fn(data)
        # inpainting re-draws the fragments around the given atoms: no templates (lightning.py:411-414)
        template_data = data if self.inpainting else create_templates_for_linker_generation(data, linker_sizes)

        x = template_data['positions']
        node_mask = template_data['atom_mask']
        edge_mask = template_data['edge_mask']
        h = template_data['one_hot']
        anchors = template_data['anchors']
        fragment_mask = template_data['fragment_mask']
        linker_mask = template_data['linker_mask']

        if self.anchors_context:
            context = torch.cat([anchors, fragment_mask], dim=-1)
        else:
            context = fragment_mask
        if self.pockets:
            fragment_only_mask = template_data['fragment_only_mask']
            pocket_only_mask = fragment_mask - fragment_only_mask
            if self.anchors_context:
                context = torch.cat([anchors, fragment_only_mask, pocket_only_mask], dim=-1)
            else:
                context = torch.cat([fragment_only_mask, pocket_only_mask], dim=-1)

        if self.inpainting:
            center_of_mass_mask = node_mask
        elif isinstance(self.val_dataset, MOADDataset) and self.center_of_mass == 'fragments':     # lightning.py:443
            center_of_mass_mask = template_data['fragment_only_mask']
        elif self.center_of_mass == 'fragments':
            center_of_mass_mask = fragment_mask
        elif self.center_of_mass == 'anchors':
            center_of_mass_mask = anchors
        else:
            raise NotImplementedError(self.center_of_mass)
        x = utils.remove_partial_mean_with_mask(x, node_mask, center_of_mass_mask)

        chain = self.edm.sample_chain(
            x=x, h=h, node_mask=node_mask, edge_mask=edge_mask, fragment_mask=fragment_mask,
            linker_mask=linker_mask, context=context, keep_frames=keep_frames,
        )
        return chain, node_mask
