"""Constants of the sampling boundary (reference ``src/const.py:6-61``; values restated,
the RDKit-bound bond tables of the reference are post-processing and out of scope)."""
import torch

TORCH_FLOAT = torch.float32
TORCH_INT = torch.int8          # masks are int8 end to end (const.py:7) — see datasets.collate

# one-hot atom vocabularies (const.py:14,29)
ATOM2IDX = {'C': 0, 'O': 1, 'N': 2, 'F': 3, 'S': 4, 'Cl': 5, 'Br': 6, 'I': 7}
IDX2ATOM = {v: k for k, v in ATOM2IDX.items()}
CHARGES = {'C': 6, 'O': 8, 'N': 7, 'F': 9, 'S': 16, 'Cl': 17, 'Br': 35, 'I': 53}
NUMBER_OF_ATOM_TYPES = len(ATOM2IDX)

GEOM_ATOM2IDX = dict(ATOM2IDX, P=8)
GEOM_IDX2ATOM = {v: k for k, v in GEOM_ATOM2IDX.items()}
GEOM_CHARGES = dict(CHARGES, P=15)
GEOM_NUMBER_OF_ATOM_TYPES = len(GEOM_ATOM2IDX)

# batch-key sets (const.py:39-47)
DATA_LIST_ATTRS = {'uuid', 'name', 'fragments_smi', 'linker_smi', 'num_atoms'}
DATA_ATTRS_TO_PAD = {
    'positions', 'one_hot', 'charges', 'anchors', 'fragment_mask', 'linker_mask', 'pocket_mask', 'fragment_only_mask'
}
DATA_ATTRS_TO_ADD_LAST_DIM = {
    'charges', 'anchors', 'fragment_mask', 'linker_mask', 'pocket_mask', 'fragment_only_mask'
}

# linker-size histogram of the ZINC train split (const.py:50-61)
# (insertion order as in the reference: DistributionNodes enumerates the dict, so the order decides which size a
# seeded Categorical draw maps to)
LINKER_SIZE_DIST = {4: 85540, 3: 113928, 6: 70946, 7: 30408, 5: 77671, 9: 5177, 10: 1214, 8: 12712, 11: 158, 12: 7}

# class tables of the size predictor (const.py:181-206)
ZINC_TRAIN_LINKER_ID2SIZE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
ZINC_TRAIN_LINKER_SIZE2ID = {size: idx for idx, size in enumerate(ZINC_TRAIN_LINKER_ID2SIZE)}
GEOM_TRAIN_LINKER_ID2SIZE = list(range(3, 33)) + [36, 38, 41]
GEOM_TRAIN_LINKER_SIZE2ID = {size: idx for idx, size in enumerate(GEOM_TRAIN_LINKER_ID2SIZE)}
