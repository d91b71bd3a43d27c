"""MultiModalLitModel with the reference's surface (reference multimodal/multimodal_lit.py:35-542).

training_step / calculate_joint_loss / configure_optimizers / encode_* / tokenize / validation trial step for
the contrastive objective (lambda_mm); the language-model and text-generation branches (lambda_lm,
eval_textgen) are outside the hot path and raise.  Logging keeps the reference's metric names."""
from __future__ import annotations

import functools
import json
import os

import numpy as np
import torch

from . import ops, parallel
from .lightning import LightningModule
from .multimodal import LanguageModel, MultiModalModel
from .multimodal_data_module import (EOS_TOKEN_ID, MAX_LEN_UTTERANCE, N_VAL_DATALOADERS_PER_SPLIT, PAD_TOKEN_ID,
                                     SOS_TOKEN_ID)
from .utils import get_entropy

OPTIMIZER = torch.optim.AdamW
LR = 3e-4
FACTOR = 0.1
PATIENCE = 20
WEIGHT_DECAY = 0.01
BEAM_WIDTH = 3
DECODE_LENGTH = MAX_LEN_UTTERANCE
LENGTH_PENALTY_ALPHA = 0.0


def _whitespace_nlp(text):
    """Tokeniser fallback when spaCy is absent: whitespace split with the Doc/Token ``.text`` shape."""
    import types
    return [types.SimpleNamespace(text=t) for t in text.split()]


class MultiModalLitModel(LightningModule):
    def __init__(self, vision_encoder, text_encoder, args):
        super().__init__()
        self.args = vars(args) if args is not None else {}
        self.optimizer_class = self.args.get("optimizer", OPTIMIZER)
        self.lr = self.args.get("lr", LR)
        self.lr_scheduler = self.args.get("lr_scheduler", False)
        self.factor = self.args.get("factor", FACTOR)
        self.patience = self.args.get("patience", PATIENCE)
        self.weight_decay = self.args.get("weight_decay", WEIGHT_DECAY)
        self.lambda_mm = self.args.get("lambda_mm", 1.)
        self.lambda_lm = self.args.get("lambda_lm", 0.)
        self.lambda_ar = self.args.get("lambda_ar", 0.)
        self.optimize_unused = self.args.get("optimize_unused", False)
        self.eval_textgen = self.args.get("eval_textgen", False)
        self.vision_encoder = vision_encoder
        self.text_encoder = text_encoder
        self.model = MultiModalModel(self.vision_encoder, self.text_encoder, args)
        self.language_model = LanguageModel(self.text_encoder, args)
        self.vocab_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vocab.json")
        with open(self.vocab_path) as f:
            self.vocab = json.load(f)
        try:                                             # reference :71
            import spacy
            self.nlp = spacy.load("en_core_web_sm")
        except Exception:
            self.nlp = _whitespace_nlp
        self.save_hyperparameters()

    @staticmethod
    def add_to_argparse(parser):
        parser.add_argument("--optimizer", type=lambda o: getattr(torch.optim, o), default=OPTIMIZER)
        parser.add_argument("--lr", type=float, default=LR)
        parser.add_argument("--lr_scheduler", action="store_true")
        parser.add_argument("--factor", type=float, default=FACTOR)
        parser.add_argument("--patience", type=int, default=PATIENCE)
        parser.add_argument("--weight_decay", type=float, default=WEIGHT_DECAY)
        parser.add_argument("--lambda_mm", type=float, default=1.)
        parser.add_argument("--lambda_lm", type=float, default=0.)
        parser.add_argument("--lambda_ar", type=float, default=0.)
        parser.add_argument("--optimize_unused", action="store_true")
        parser.add_argument("--eval_textgen", action="store_true")
        parser.add_argument("--beam_width", type=int, default=BEAM_WIDTH)
        parser.add_argument("--decode_length", type=int, default=DECODE_LENGTH)
        parser.add_argument("--length_penalty_alpha", type=float, default=LENGTH_PENALTY_ALPHA)

    def set_precision(self, precision):
        """Trainer ``--precision``: 'bf16' / '16' -> bf16 storage + bf16 MFMA trunk; '32' -> exact-fp32 parity mode;
        'fp8' -> bf16 storage with e4m3 weights / activations in the ViT linears (BASELINE configs[4]; the ResNeXt trunk
        has no fp8 path and runs in bf16)."""
        p = str(precision)
        dt = torch.bfloat16 if p in ("bf16", "16", "bf16-mixed", "16-mixed", "fp8", "8") else torch.float32
        self.vision_encoder.set_compute_dtype(dt)
        # the text transformer's fp32 linears: exact-fp32 MFMA in the parity mode, hi / lo split bf16 MFMA (~2^-16) otherwise
        self.text_encoder.__dict__["fp32_split"] = dt == torch.bfloat16
        if getattr(self.vision_encoder, "vit_dino", False):
            self.vision_encoder.model.fp8_linears = p in ("fp8", "8")

    # ---- Lightning checkpoint hooks: the ResNeXt trunk's bf16 storage centres (resnext.py "centred storage") ----------------
    # The centres are calibrated on the first train-mode batch and are NOT parameters or buffers (the state_dict keeps the
    # reference's / torchvision's key set), so a resumed run would recalibrate on a different batch and stop being bit-reproducible.
    # They travel in the checkpoint under their own top-level key; a checkpoint without the key (the reference's) simply recalibrates.
    def on_save_checkpoint(self, checkpoint):
        trunk = getattr(self.vision_encoder, "model", None)
        trunk = getattr(trunk, "_resnet", trunk)
        exp = getattr(trunk, "export_centres", None)
        st = exp() if exp is not None else None
        if st:
            checkpoint["cvcl_storage_centres"] = st

    def on_load_checkpoint(self, checkpoint):
        trunk = getattr(self.vision_encoder, "model", None)
        trunk = getattr(trunk, "_resnet", trunk)
        st = checkpoint.get("cvcl_storage_centres")
        if st and hasattr(trunk, "import_centres"):
            trunk.import_centres(st)

    def configure_optimizers(self):
        kw = {}
        params = list(self.parameters())
        if self.optimizer_class in (torch.optim.AdamW, torch.optim.Adam) and params and all(p.is_cuda for p in params):
            kw["fused"] = True           # same update rule in one multi-tensor launch instead of ~8 (PyTorch is the optimizer per north_star)
        optimizer = self.optimizer_class(params, lr=self.lr, weight_decay=self.weight_decay, **kw)
        if not self.lr_scheduler:
            return optimizer
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, factor=self.factor, patience=self.patience)
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": sched, "monitor": "val_loss"}}

    def forward(self, x, y, y_len):
        return self.model(x, y, y_len)

    @staticmethod
    def load_model(model_name="cvcl", checkpoint_path=None):
        """Reference :134-149 downloads ``wkvong/cvcl_s_dino_resnext50_embedding`` from the HF hub; without a
        network the checkpoint path must be given (or $CVCL_CHECKPOINT)."""
        if model_name != "cvcl":
            raise ValueError("Model name not found.")
        path = checkpoint_path or os.environ.get("CVCL_CHECKPOINT")
        if not path or not os.path.isfile(path):
            raise FileNotFoundError("no network access: pass checkpoint_path or set CVCL_CHECKPOINT to a local "
                                    "cvcl_s_dino_resnext50_embedding.ckpt")
        model = MultiModalLitModel.load_from_checkpoint(checkpoint_path=path)
        return model, None

    def encode_image(self, x):
        return self.model.encode_image(x)[0]

    def encode_text(self, y, y_len=None):
        return self.model.encode_text(y, y_len)[0]

    def tokenize(self, texts):
        """``<sos> tokens <eos>`` padded to 25, length = n + 2 (reference :161-190)."""
        max_seq_len = 25
        if isinstance(texts, str):
            texts = [texts]
        all_tokens, lengths = [], []
        for text in texts:
            words = [t.text for t in self.nlp(text)][:max_seq_len - 2]
            ids = [self.vocab["<sos>"]] + [self.vocab.get(w, self.vocab["<unk>"]) for w in words] + [self.vocab["<eos>"]]
            lengths.append(len(ids))
            all_tokens.append(ids + [self.vocab["<pad>"]] * (max_seq_len - len(ids)))
        return torch.tensor(all_tokens, dtype=torch.long), torch.tensor(lengths, dtype=torch.long)

    def calculate_ce_loss(self, y, y_len, x=None, outputs=None, image_features=None, image_feature_map=None,
                          return_image_features=False, **kwargs):
        """Wraps language_model.calculate_ce_loss (reference :192-225; captioning / attention variants out of scope)."""
        te = self.language_model.text_encoder
        if te.captioning or te.has_attention:
            raise NotImplementedError("captioning / attention language models are outside the implemented path")
        ret = self.language_model.calculate_ce_loss(y, y_len, outputs=outputs, **kwargs)
        if return_image_features:
            ret = ret + (None, None)
        return ret

    def calculate_joint_loss(self, batch, stage, log, eval_textgen=False, ce_weight=None):
        x, y, y_len, raw_y = batch
        ret = {"batch_size": x.size(0)}
        if self.lambda_mm or not self.optimize_unused:
            infonce_loss, image_accuracy, text_accuracy, image_entropy, text_entropy, *_rest = \
                self.model.calculate_contrastive_loss(x, y, y_len)
            log(f"{stage}_infonce_loss", infonce_loss)
            log(f"{stage}_image_accuracy", image_accuracy)
            log(f"{stage}_text_accuracy", text_accuracy)
            log(f"{stage}_image_entropy", image_entropy)
            log(f"{stage}_text_entropy", text_entropy)
            # reference :252-253 calls .item() here (a host sync per step); the tensor is logged instead
            log("temperature", (-self.model.logit_neg_log_temperature.detach()).exp())
            ret.update({"infonce_loss": infonce_loss.detach(), "image_accuracy": image_accuracy,
                        "text_accuracy": text_accuracy, "image_entropy": image_entropy.detach(),
                        "text_entropy": text_entropy.detach()})
            text_outputs = _rest[-1]
        else:
            infonce_loss = 0.
            text_outputs = None
        if self.lambda_lm or not self.optimize_unused:                       # reference :266-309
            if eval_textgen:
                raise NotImplementedError("text generation evaluation (beam search) is outside the implemented path")
            ce_loss, _o, _l, _attns, labels = self.calculate_ce_loss(y, y_len, x=x, outputs=text_outputs, tokenwise=True,
                                                                      weight=ce_weight)
            means, counts = ops.lm_loss_summaries(ce_loss.reshape(-1), labels.reshape(-1), PAD_TOKEN_ID, SOS_TOKEN_ID, EOS_TOKEN_ID)
            lm_ce_loss = means[0]
            for i, suffix in enumerate(("", "_wo_sos", "_wo_sos_eos")):
                log(f"{stage}_ce_loss{suffix}", means[i].detach())
                ret[f"ce_loss{suffix}"] = means[i].detach()
                ret[f"n_tokens{suffix}"] = counts[i]
        else:
            lm_ce_loss = 0.
        # data-parallel, global negatives: InfoNCE is the replicated full-batch loss and per-rank gradients are SUMMED; the LM
        # cross entropy is a mean over this rank's tokens only, so it enters with 1 / world: the sum over ranks is then the mean
        # of the rank means -- the token-weighted global mean a single process computes whenever the ranks hold the same number
        # of tokens (equal-length shards), an approximation of it otherwise (like the mean of per-rank means under Lightning
        # DDP).  The optimised loss carries the scale; the logged / returned values are the unscaled joint loss and the rank's
        # own ce means.
        lm_scale = parallel.local_term_scale(self.model.global_negatives) if (self.training and self.lambda_lm) else 1.0
        if isinstance(lm_ce_loss, float) and torch.is_tensor(infonce_loss):
            # contrastive term only (lambda_lm = 0, the BASELINE configurations): no "+ 0." / "1.0 *" elementwise launches
            loss = infonce_loss if self.lambda_mm == 1 else self.lambda_mm * infonce_loss
        else:
            loss = self.lambda_mm * infonce_loss + (self.lambda_lm * lm_scale) * lm_ce_loss
        logged = loss if lm_scale == 1.0 else (self.lambda_mm * infonce_loss + self.lambda_lm * lm_ce_loss).detach()
        log(f"{stage}_loss", logged)
        ret.update({"loss": loss})
        return ret

    def joint_loss_epoch_end(self, outputs, stage, log, eval_textgen=False):
        def mean_over_examples(name):
            n, total = 0, 0.
            for o in outputs:
                n += o["batch_size"]
                total += float(o[name]) * o["batch_size"]
            return total / n
        def mean_over_tokens(name, n_name):
            n, total = 0., 0.
            for o in outputs:
                n += float(o[n_name])
                total += float(o[name]) * float(o[n_name])
            return total / n
        if self.lambda_mm or not self.optimize_unused:
            for name in ("infonce_loss", "image_accuracy", "text_accuracy", "image_entropy", "text_entropy"):
                log(f"{stage}_{name}", mean_over_examples(name))
        if self.lambda_lm or not self.optimize_unused:                       # reference :420-428
            for suffix in ("", "_wo_sos", "_wo_sos_eos"):
                value_mean = mean_over_tokens(f"ce_loss{suffix}", f"n_tokens{suffix}")
                log(f"{stage}_ce_loss{suffix}", value_mean)
                log(f"{stage}_perplexity{suffix}", float(np.exp(value_mean)))
        log(f"{stage}_loss", mean_over_examples("loss"))

    def training_step(self, batch, batch_idx):
        return self.calculate_joint_loss(batch, "train", self.log, eval_textgen=False)

    def training_epoch_end(self, outputs):
        log = lambda name, value, *a, **k: self.log(f"{name}_epoch", value, on_step=False, on_epoch=True, *a, **k)
        return self.joint_loss_epoch_end(outputs, "train", log, eval_textgen=False)

    def validation_test_step(self, stage, batch, batch_idx, dataloader_idx=0):
        log = functools.partial(self.log, on_step=False, on_epoch=True)
        ret = {}
        if dataloader_idx == 0:
            ret.update(self.calculate_joint_loss(batch, stage, lambda *a, **k: None, eval_textgen=self.eval_textgen))
        elif dataloader_idx == 1:                        # one 4-way trial per batch (reference :466-511)
            x, y, y_len, raw_y = batch
            x = x.view(-1, *x.shape[-3:])
            if self.lambda_mm:
                logits_per_image, logits_per_text = self.model(x, y, y_len)
                logits = logits_per_text[0]
                pred = torch.argmax(logits).item()
                accuracy = int(pred == 0)
                log(f"{stage}_accuracy", accuracy)
                log(f"{stage}_entropy", get_entropy(logits))
                log(f"{stage}_accuracy_{raw_y[0][0]}", accuracy)
                ret.update({"accuracy": accuracy})
        return ret

    def validation_test_epoch_end(self, stage, outputs):
        log = functools.partial(self.log, on_step=False, on_epoch=True)
        return self.joint_loss_epoch_end(outputs[0], stage, log, eval_textgen=self.eval_textgen)

    def validation_step(self, batch, batch_idx, dataloader_idx=0):
        if dataloader_idx < N_VAL_DATALOADERS_PER_SPLIT:
            return self.validation_test_step("val", batch, batch_idx, dataloader_idx=dataloader_idx)
        return self.test_step(batch, batch_idx, dataloader_idx=dataloader_idx - N_VAL_DATALOADERS_PER_SPLIT)

    def validation_epoch_end(self, outputs):
        self.validation_test_epoch_end("val", outputs[:N_VAL_DATALOADERS_PER_SPLIT])

    def test_step(self, batch, batch_idx, dataloader_idx=0):
        return self.validation_test_step("test", batch, batch_idx, dataloader_idx=dataloader_idx)

    def test_epoch_end(self, outputs):
        return self.validation_test_epoch_end("test", outputs)
