"""Batch contract of the reference data modules + a synthetic data module.

Mirrors the constants, ``multiModalDataset_collate_fn`` and CLI flags of the reference
(multimodal/multimodal_data_module.py:26-54, 98-109, 283-311; multimodal_saycam_data_module.py:93-124,
142-150).  The SAYCam / COCO loaders themselves read a private dataset from hard-coded cluster paths and
are out of scope; ``SyntheticDataModule`` produces batches of the same shape."""
from __future__ import annotations

import json
import os

import torch
from torch.nn.utils.rnn import pad_sequence

from .lightning import LightningDataModule

BATCH_SIZE = 4
VAL_BATCH_SIZE = 16
NUM_WORKERS = 4
EVAL_INCLUDE_SOS_EOS = False
N_VAL_DATALOADERS_PER_SPLIT = 2
TEST_WHILE_VAL = False
EVAL_TYPE = "image"
MAX_LEN_UTTERANCE = 25
AUGMENT_FRAMES = False
PAD_TOKEN, UNK_TOKEN, SOS_TOKEN, EOS_TOKEN = "<pad>", "<unk>", "<sos>", "<eos>"
PAD_TOKEN_ID, UNK_TOKEN_ID, SOS_TOKEN_ID, EOS_TOKEN_ID = 0, 1, 2, 3
IMAGE_H = IMAGE_W = 224
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
CLIP_EVAL = False
VOCAB_FILENAME = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vocab.json")


def read_vocab(vocab_filename=VOCAB_FILENAME):
    with open(vocab_filename) as f:
        return json.load(f)


def multiModalDataset_collate_fn(batch):
    """(img, idxs, len, raw) items -> (img [B,3,H,W], idxs [B,Lmax<=25] pad 0, len [B] int64, raw list)."""
    img, idxs, length, raw = zip(*batch)
    img = torch.stack(img, 0)
    idxs = pad_sequence(idxs, batch_first=True, padding_value=PAD_TOKEN_ID)
    length = torch.tensor(length, dtype=torch.long)
    if idxs.size(1) > MAX_LEN_UTTERANCE:
        idxs = idxs[:, :MAX_LEN_UTTERANCE]
        length = torch.clamp(length, max=MAX_LEN_UTTERANCE)
    return img, idxs, length, list(raw)


class MultiModalDataModule(LightningDataModule):
    def __init__(self, args=None):
        super().__init__()
        self.args = vars(args) if args is not None else {}
        self.batch_size = self.args.get("batch_size", BATCH_SIZE)
        self.drop_last = self.args.get("drop_last", False)
        self.val_batch_size = self.args.get("val_batch_size", VAL_BATCH_SIZE)
        self.num_workers = self.args.get("num_workers", NUM_WORKERS)
        self.augment_frames = self.args.get("augment_frames", False)
        # --device_frames: the datasets hand over decoded uint8 frames [H, W, 3] and the transform of :244-256 (or the base
        # transform) runs on the device after the batch transfer (multimodal/augment.py) instead of per frame in the workers
        self.device_frames = bool(self.args.get("device_frames", False))
        self._frame_transforms = None

    def on_after_batch_transfer(self, batch, dataloader_idx=0, training=True):
        """Lightning's hook of the same name: uint8 frame batches [B, H, W, 3] (or evaluation trials [B, n, H, W, 3]) become
        normalised fp32 [.., 3, 224, 224] tensors on the device; the training transform only while training (the reference
        keeps ``base_transform`` for val / test, :271-275).  Batches that already hold float images pass through."""
        img = batch[0] if isinstance(batch, (list, tuple)) and len(batch) > 0 else None
        if not torch.is_tensor(img) or img.dtype != torch.uint8:
            return batch
        from .augment import DeviceFrameAugment
        if self._frame_transforms is None:
            self._frame_transforms = {True: DeviceFrameAugment(augment_frames=self.augment_frames),
                                      False: DeviceFrameAugment(augment_frames=False)}
        tf = self._frame_transforms[bool(training)]
        lead = img.shape[:-3]
        out = tf(img.reshape(-1, *img.shape[-3:]))
        return type(batch)((out.reshape(*lead, *out.shape[1:]),) + tuple(batch[1:]))

    @staticmethod
    def add_to_argparse(parser):
        parser.add_argument("--batch_size", type=int, default=BATCH_SIZE)
        parser.add_argument("--drop_last", action="store_true")
        parser.add_argument("--val_batch_size", type=int, default=VAL_BATCH_SIZE)
        parser.add_argument("--num_workers", type=int, default=NUM_WORKERS)
        parser.add_argument("--augment_frames", action="store_true")
        parser.add_argument("--device_frames", action="store_true",
                            help="datasets yield uint8 frames; the (augmentation or base) transform runs on the GPU per batch")
        parser.add_argument("--eval_include_sos_eos", action="store_true")
        parser.add_argument("--test_while_val", action="store_true")
        parser.add_argument("--eval_type", type=str, default=EVAL_TYPE, choices=["image", "text"])
        parser.add_argument("--eval_metadata_filename", type=str, default="eval_filtered_dev.json")
        parser.add_argument("--clip_eval", action="store_true")
        return parser

    @staticmethod
    def add_additional_to_argparse(parser):
        parser.add_argument("--multiple_frames", action="store_true")
        parser.add_argument("--shuffle_utterances", action="store_true")
        parser.add_argument("--multiple_captions", action="store_true")
        return parser

    def read_vocab(self):
        return read_vocab()


class SyntheticPairs(torch.utils.data.Dataset):
    """``rand -> ImageNet normalise`` frames and ``<sos> w1..wn <eos>`` utterances (reference item shape:
    multimodal_saycam_data_module.py:93-124; image statistics: multimodal_data_module.py:57)."""

    def __init__(self, n_items: int, vocab_size: int, n_words: int = 3, seed: int = 0, raw_frames: bool = False):
        self.n, self.vocab_size, self.n_words, self.seed = n_items, vocab_size, n_words, seed
        self.raw_frames = raw_frames                       # uint8 [H, W, 3] frames for the device transform
        self.mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
        self.std = torch.tensor(IMAGENET_STD).view(3, 1, 1)

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        if self.raw_frames:
            img = torch.randint(0, 256, (IMAGE_H, IMAGE_W, 3), dtype=torch.uint8, generator=g)
        else:
            img = (torch.rand(3, IMAGE_H, IMAGE_W, generator=g) - self.mean) / self.std
        words = torch.randint(4, self.vocab_size, (self.n_words,), generator=g)
        idxs = torch.cat([torch.tensor([SOS_TOKEN_ID]), words, torch.tensor([EOS_TOKEN_ID])]).long()
        return img, idxs, int(idxs.numel()), [" ".join(f"w{int(w)}" for w in words)]


class SyntheticEvalTrials(torch.utils.data.Dataset):
    """4-way evaluation trials with the item layout of the reference's LabeledSEvalDataset (``eval_type='image'``:
    (imgs [4,3,H,W] target first, label ids [L], label length, [raw category]); multimodal_data_module.py:112-160) or
    LabeledSTextEvalDataset (``eval_type='text'``: (img [1,3,H,W], label ids [4,L] target first, [4 lengths], [raw target]);
    :163-213).  ``metadata()`` is the list the reference reads from eval_*.json (target_category / foil_categories)."""

    def __init__(self, n_trials, vocab_size, seed=0, eval_include_sos_eos=False, n_images=4, raw_frames=False, eval_type="image"):
        self.n, self.v, self.seed, self.sos_eos, self.n_images = n_trials, vocab_size, seed, eval_include_sos_eos, n_images
        self.raw_frames = raw_frames
        self.eval_type = eval_type

    def __len__(self):
        return self.n

    def _words(self, idx):
        g = torch.Generator().manual_seed(self.seed * 7919 + idx)
        words = torch.randperm(self.v - 4, generator=g)[: self.n_images] + 4          # target + distinct foil categories
        return g, [int(w) for w in words]

    def metadata(self):
        out = []
        for idx in range(self.n):
            _, words = self._words(idx)
            out.append({"target_category": f"w{words[0]}", "foil_categories": [f"w{w}" for w in words[1:]]})
        return out

    def _wrap(self, word):
        return [SOS_TOKEN_ID, word, EOS_TOKEN_ID] if self.sos_eos else [word]

    def __getitem__(self, idx):
        g, words = self._words(idx)
        n_img = self.n_images if self.eval_type == "image" else 1
        if self.raw_frames:
            imgs = torch.randint(0, 256, (n_img, IMAGE_H, IMAGE_W, 3), dtype=torch.uint8, generator=g)
        else:
            mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
            std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
            imgs = (torch.rand(n_img, 3, IMAGE_H, IMAGE_W, generator=g) - mean) / std
        if self.eval_type == "image":
            label = self._wrap(words[0])
            return imgs, torch.tensor(label, dtype=torch.long), len(label), [f"w{words[0]}"]
        labels = [self._wrap(w) for w in words]
        return imgs, torch.tensor(labels, dtype=torch.long), [len(l) for l in labels], [f"w{words[0]}"]


class SyntheticDataModule(MultiModalDataModule):
    """``--dataset synthetic``: same batch contract as the SAYCam module, no files needed."""

    def __init__(self, args=None, n_items: int = 64):
        super().__init__(args)
        world = int(os.environ.get("WORLD_SIZE", "1"))
        self.n_items = max(n_items, 2 * self.batch_size) * max(world, 1)          # the same number of steps per rank
        self.seed = self.args.get("seed", 0)

    def prepare_data(self, *a, **k):
        pass

    def setup(self, *a, **k):
        v = len(self.read_vocab())
        raw = self.device_frames
        self.train_set = SyntheticPairs(self.n_items, v, seed=self.seed, raw_frames=raw)
        self.val_set = SyntheticPairs(self.val_batch_size, v, seed=self.seed + 1, raw_frames=raw)
        self.test_set = SyntheticPairs(self.val_batch_size, v, seed=self.seed + 2, raw_frames=raw)
        sos_eos = bool(self.args.get("eval_include_sos_eos", False))
        et = self.args.get("eval_type", EVAL_TYPE) or EVAL_TYPE
        n_trials = int(self.args.get("n_eval_trials", 4) or 4)
        self.eval_sets = {"val": SyntheticEvalTrials(n_trials, v, seed=self.seed + 3, eval_include_sos_eos=sos_eos, raw_frames=raw,
                                                     eval_type=et),
                          "test": SyntheticEvalTrials(n_trials, v, seed=self.seed + 4, eval_include_sos_eos=sos_eos, raw_frames=raw,
                                                      eval_type=et)}

    def set_epoch(self, epoch: int):
        self._epoch = int(epoch)

    def train_dataloader(self):
        """One process per GPU: every rank reads its own shard of each global batch (DistributedSampler, as Lightning's DDP
        strategy injects into the reference's loaders) -- identical batches on every rank would make each positive a
        ``world``-fold negative of itself under global negatives."""
        sampler = None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(
                self.train_set, num_replicas=torch.distributed.get_world_size(), rank=torch.distributed.get_rank(),
                shuffle=False, drop_last=True)
            sampler.set_epoch(getattr(self, "_epoch", 0))
        return torch.utils.data.DataLoader(self.train_set, batch_size=self.batch_size, shuffle=False, sampler=sampler,
                                           collate_fn=multiModalDataset_collate_fn, drop_last=self.drop_last)

    def _val_test(self, pairs, trials):
        """reference val_test_dataloader (:378-403): [pair batches, one evaluation trial per batch]"""
        return [torch.utils.data.DataLoader(pairs, batch_size=self.val_batch_size, shuffle=False,
                                            collate_fn=multiModalDataset_collate_fn),
                torch.utils.data.DataLoader(trials, batch_size=1, shuffle=False, collate_fn=multiModalDataset_collate_fn)]

    def val_dataloader(self):
        return self._val_test(self.val_set, self.eval_sets["val"])

    def test_dataloader(self):
        return self._val_test(self.test_set, self.eval_sets["test"])
