"""The slice of pytorch_lightning 1.6 that train.py and MultiModalLitModel rely on.

pytorch_lightning is not installed in this image, and the reference pins 1.6.0 whose Trainer predates
ROCm 7 wheels; this module provides ``LightningModule`` / ``LightningDataModule`` / ``Trainer`` /
``seed_everything`` with the behaviours the entry point uses: the Trainer CLI flags, ``--fast_dev_run``,
the fit loop (train()/eval(), batch to device, training_step -> backward -> optimizer.step), self.log
collection, ``save_hyperparameters`` and a one-process-per-GPU data-parallel mode over RCCL."""
from __future__ import annotations

import argparse
import contextlib
import os
import random

import numpy as np
import torch
import torch.nn as nn


def seed_everything(seed: int):
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    os.environ["PL_GLOBAL_SEED"] = str(seed)
    return seed


def _str2bool(v):
    if isinstance(v, bool):
        return v
    return str(v).lower() in ("1", "true", "yes", "y")


class LightningDataModule:
    def __init__(self):
        pass

    def prepare_data(self, *a, **k):
        pass

    def setup(self, *a, **k):
        pass


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._logged = {}
        self.hparams = {}
        self.trainer = None

    def save_hyperparameters(self, *args, **kwargs):
        """Record the constructor arguments of the calling ``__init__`` (what Lightning stores as hyper_parameters)."""
        import inspect
        local = inspect.currentframe().f_back.f_locals
        names = [n for n in inspect.signature(type(self).__init__).parameters if n != "self"]
        self.hparams = {k: local[k] for k in names if k in local}

    def log(self, name, value, *args, **kwargs):
        """Last value wins for step-level logs; ``on_step=False, on_epoch=True`` logs (the evaluation loops) are
        averaged over the epoch like Lightning's default mean reduction."""
        if kwargs.get("on_epoch") and kwargs.get("on_step") is False:
            acc = self.__dict__.setdefault("_log_accum", {}).setdefault(name, [0.0, 0])
            acc[0] += float(value)
            acc[1] += 1
            self._logged[name] = acc[0] / acc[1]
        else:
            self._logged[name] = value

    def _reset_epoch_logs(self):
        self.__dict__["_log_accum"] = {}

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    def configure_optimizers(self):
        raise NotImplementedError

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, **kwargs):
        """Lightning ckpt layout: ``state_dict`` + ``hyper_parameters`` (the constructor arguments, which for
        MultiModalLitModel include the pickled encoder modules: reference multimodal_lit.py:74,139)."""
        from . import resnext
        resnext.register_torchvision_alias()
        ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False)
        hp = dict(ckpt["hyper_parameters"])
        hp.update(kwargs)
        model = cls(**hp)
        model.load_state_dict(ckpt["state_dict"])
        model.on_load_checkpoint(ckpt)
        return model

    def on_save_checkpoint(self, checkpoint):                 # Lightning's hooks: extra (non-state_dict) state of a module
        pass

    def on_load_checkpoint(self, checkpoint):
        pass

    def checkpoint_dict(self, trainer=None, optimizer=None, scheduler=None):
        """Lightning's checkpoint keys: weights, hyper-parameters, loop position, optimizer / lr-scheduler / callback state."""
        callbacks = {type(cb).__name__: cb.state_dict() for cb in getattr(trainer, "callbacks", []) if hasattr(cb, "state_dict")}
        ckpt = {"state_dict": self.state_dict(), "hyper_parameters": dict(self.hparams),
                "epoch": getattr(trainer, "current_epoch", 0), "global_step": getattr(trainer, "global_step", 0),
                "optimizer_states": [optimizer.state_dict()] if optimizer is not None else [],
                "lr_schedulers": [scheduler.state_dict()] if scheduler is not None else [],
                "callbacks": callbacks,
                "pytorch-lightning_version": "1.6.0-shim"}
        self.on_save_checkpoint(ckpt)
        return ckpt


def _move(batch, device):
    if torch.is_tensor(batch):
        return batch.to(device, non_blocking=True)
    if isinstance(batch, (list, tuple)):
        return type(batch)(_move(b, device) for b in batch)
    return batch


class Trainer:
    """Minimal fit loop with the flags the reference's configs use."""

    FLAGS = (("gpus", int, 0), ("max_epochs", int, 1), ("check_val_every_n_epoch", int, 1),
             ("checkpoint_callback", _str2bool, True), ("logger", _str2bool, True), ("fast_dev_run", None, False),
             ("strategy", str, None), ("devices", int, None), ("precision", str, "32"), ("limit_train_batches", int, None),
             ("limit_val_batches", int, None), ("trunk_stream", _str2bool, False))

    @classmethod
    def add_argparse_args(cls, parser):
        for name, typ, default in cls.FLAGS:
            if name == "fast_dev_run":
                parser.add_argument("--fast_dev_run", nargs="?", const=True, default=False, type=_str2bool)
            else:
                parser.add_argument("--" + name, type=typ, default=default)
        return parser

    @classmethod
    def from_argparse_args(cls, args, **kwargs):
        kw = {name: getattr(args, name, default) for name, _t, default in cls.FLAGS}
        kw.update(kwargs)
        return cls(**kw)

    def __init__(self, gpus=0, max_epochs=1, check_val_every_n_epoch=1, checkpoint_callback=True, logger=True,
                 fast_dev_run=False, strategy=None, devices=None, precision="32", limit_train_batches=None,
                 limit_val_batches=None, callbacks=None, enable_checkpointing=None, trunk_stream=False, **_ignored):
        self.gpus, self.max_epochs = gpus or 0, max_epochs
        self.fast_dev_run = fast_dev_run
        # --trunk_stream True: a frozen image trunk, the host-to-device copy of the batch and the device frame transform run on
        # their own HIP stream and overlap the previous step's trainable tail (H.TrunkStream); same numbers, bit for bit
        self.trunk_stream = bool(trunk_stream)
        self.precision = str(precision)
        self.limit_train_batches = 1 if fast_dev_run else limit_train_batches
        self.limit_val_batches = 1 if fast_dev_run else limit_val_batches
        self.check_val_every_n_epoch = max(int(check_val_every_n_epoch or 1), 1)
        if fast_dev_run:
            self.max_epochs = 1
        self.callbacks = callbacks or []
        self.logged_metrics = {}
        self.global_step = 0
        self.current_epoch = 0
        # `--checkpoint_callback` is the pre-1.7 alias of enable_checkpointing (reference train.py:97)
        self.enable_checkpointing = bool(checkpoint_callback if enable_checkpointing is None else enable_checkpointing) \
            and not fast_dev_run

    def _device(self):
        if self.gpus and self.gpus > 0:
            if not torch.cuda.is_available():
                raise RuntimeError("--gpus > 0 but no GPU is visible")
            # one process per GPU; the modulo only matters when more ranks than GPUs are started on purpose
            # (CVCL_DIST_BACKEND=gloo smoke runs of the N > 1 path on a single-GPU box)
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1))
            torch.cuda.set_device(device)      # libcvcl_hip launches on the CURRENT device's current stream: rank r > 0 must not
            return device                      # enqueue on cuda:0 (validate()/test() never reach DataParallelEngine.from_env)
        return torch.device("cpu")

    def _eval_loop(self, model, loaders, step_name, epoch_end_name, device, datamodule=None):
        """Lightning's evaluation loop: eval mode (BatchNorm on running statistics, no dropout), no autograd, one pass
        over every dataloader with its ``dataloader_idx``, then the ``*_epoch_end`` hook on the per-loader output lists."""
        if loaders is None:
            return {}
        if not isinstance(loaders, (list, tuple)):
            loaders = [loaders]
        was_training = model.training
        model.eval()
        model._reset_epoch_logs()
        outputs = []
        with torch.no_grad():
            for di, dl in enumerate(loaders):
                outs = []
                for bi, batch in enumerate(dl):
                    if self.limit_val_batches is not None and bi >= self.limit_val_batches:
                        break
                    batch = _move(batch, device)
                    if datamodule is not None and hasattr(datamodule, "on_after_batch_transfer"):
                        batch = datamodule.on_after_batch_transfer(batch, di, training=False)
                    out = getattr(model, step_name)(batch, bi, dataloader_idx=di)
                    outs.append({k: (v.detach() if torch.is_tensor(v) else v) for k, v in (out or {}).items()})
                outputs.append(outs)
            if hasattr(model, epoch_end_name) and outputs and outputs[0]:
                getattr(model, epoch_end_name)(outputs)
        self.logged_metrics.update({k: (float(v.detach()) if torch.is_tensor(v) and v.numel() == 1 else v) for k, v in model._logged.items()})
        model.train(was_training)
        return dict(self.logged_metrics)

    def _prepare(self, model, datamodule):
        device = self._device()
        model.trainer = self
        datamodule.prepare_data()
        datamodule.setup()
        model.to(device)
        if hasattr(model, "set_precision"):
            model.set_precision(self.precision)
        return device

    def validate(self, model, datamodule=None):
        device = self._prepare(model, datamodule)
        return [self._eval_loop(model, datamodule.val_dataloader(), "validation_step", "validation_epoch_end", device, datamodule)]

    def test(self, model, datamodule=None):
        device = self._prepare(model, datamodule)
        return [self._eval_loop(model, datamodule.test_dataloader(), "test_step", "test_epoch_end", device, datamodule)]

    def fit(self, model, datamodule=None, ckpt_path=None):
        from . import parallel
        device = self._device()
        # --local_negatives: per-rank B x B loss, gradients AVERAGED (Lightning DDP); default: global negatives, gradients SUMMED
        global_negatives = bool(getattr(getattr(model, "model", None), "global_negatives", True))
        engine = parallel.DataParallelEngine.from_env(device, global_negatives=global_negatives)
        if parallel.is_distributed():
            # replicas were initialised from the same seed (identical weights); from here on every rank draws its own dropout
            # masks / augmentation parameters, and the data module hands it its own shard of every global batch
            # (seeded directly: seed_everything would overwrite $PL_GLOBAL_SEED with base + rank, and a second fit() in the same
            # process -- resume, fit after fit -- would add the rank again)
            base = int(os.environ.get("PL_GLOBAL_SEED", torch.initial_seed() % (2 ** 31)))
            random.seed(base + parallel.rank())
            np.random.seed((base + parallel.rank()) % (2 ** 32))
            torch.manual_seed(base + parallel.rank())
        model.trainer = self
        datamodule.prepare_data()
        datamodule.setup()
        model.to(device)
        if hasattr(model, "set_precision"):
            model.set_precision(self.precision)
        opt = model.configure_optimizers()
        sched = None
        if isinstance(opt, dict):
            sched, opt = opt.get("lr_scheduler"), opt["optimizer"]
        start_epoch = 0
        if ckpt_path is not None:
            if not os.path.isfile(str(ckpt_path)):
                raise FileNotFoundError(f"Checkpoint at {ckpt_path} not found. Aborting training.")     # as Lightning does
            ckpt = torch.load(str(ckpt_path), map_location="cpu", weights_only=False)
            model.load_state_dict(ckpt["state_dict"])
            model.on_load_checkpoint(ckpt)
            if ckpt.get("optimizer_states"):
                opt.load_state_dict(ckpt["optimizer_states"][0])
            if sched is not None and ckpt.get("lr_schedulers"):
                sched["scheduler"].load_state_dict(ckpt["lr_schedulers"][0])
            for cb in self.callbacks:
                st = (ckpt.get("callbacks") or {}).get(type(cb).__name__)
                if st is not None and hasattr(cb, "load_state_dict"):
                    cb.load_state_dict(st)
            start_epoch, self.global_step = int(ckpt.get("epoch", -1)) + 1, int(ckpt.get("global_step", 0))
        engine.attach(model)
        upd = None
        if parallel.is_distributed() and hasattr(model, "vision_encoder"):
            upd = parallel.OverlappedUpdate(engine, opt, model.vision_encoder)
        ts = None
        trunk = getattr(getattr(model, "vision_encoder", None), "model", None)
        if self.trunk_stream and device.type == "cuda" and hasattr(trunk, "enable_trunk_stream") \
                and not any(p.requires_grad for n, p in trunk.named_parameters() if not n.startswith(("fc.", "head."))):
            torch.cuda.synchronize(device)
            ts = trunk.enable_trunk_stream(device, inputs="caller")    # "ready" only inside the training loop (below)
        for epoch in range(start_epoch, self.max_epochs):
            self.current_epoch = epoch
            model.train()
            model._reset_epoch_logs()          # epoch means start from zero (also when validation is skipped this epoch)
            if hasattr(datamodule, "set_epoch"):
                datamodule.set_epoch(epoch)
            outs = []
            if ts is not None:
                ts.inputs = "ready"            # the batch is produced on the trunk stream itself (below)
            for bi, batch in enumerate(datamodule.train_dataloader()):
                if self.limit_train_batches is not None and bi >= self.limit_train_batches:
                    break
                with (ts.context() if ts is not None else contextlib.nullcontext()):
                    batch = _move(batch, device)
                    if hasattr(datamodule, "on_after_batch_transfer"):
                        batch = datamodule.on_after_batch_transfer(batch, 0, training=True)
                if ts is not None:            # tokens / lengths were copied on the trunk stream and are read on this one
                    main = torch.cuda.current_stream(device)
                    copied = torch.cuda.Event()
                    copied.record(ts.stream)
                    main.wait_event(copied)    # (the copy sits right behind the previous trunk: this stream is past it anyway)
                    for t in batch:
                        if torch.is_tensor(t) and t.is_cuda:
                            t.record_stream(main)
                if upd is None:
                    opt.zero_grad(set_to_none=True)
                    out = model.training_step(batch, bi)
                    out["loss"].backward()
                    engine.reduce_gradients()
                    opt.step()
                else:              # multi-GPU, frozen trunk: step k's all-reduce + update overlap step k+1's trunk forward
                    out = model.training_step(batch, bi)
                    upd.zero_grad()
                    out["loss"].backward()
                    upd.step_done()
                self.global_step += 1
                outs.append({k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()})
            if upd is not None:
                upd.flush()                    # every update applied before epoch-end hooks / validation / checkpoints
            if ts is not None:
                ts.join()                      # BatchNorm buffers written on the trunk stream are read (saved) on this one
                ts.inputs = "caller"           # evaluation batches are produced on this stream
            if hasattr(model, "training_epoch_end") and outs:
                model.training_epoch_end(outs)
            self.logged_metrics.update(model._logged)
            if (epoch + 1) % self.check_val_every_n_epoch == 0 and hasattr(datamodule, "val_dataloader") \
                    and hasattr(model, "validation_step"):
                self._eval_loop(model, datamodule.val_dataloader(), "validation_step", "validation_epoch_end", device, datamodule)
            if sched is not None and "val_loss" in self.logged_metrics:
                sched["scheduler"].step(float(self.logged_metrics["val_loss"]))
            if self.enable_checkpointing and parallel.rank() == 0:
                for cb in self.callbacks:
                    if hasattr(cb, "save"):
                        cb.save(self, model, opt, sched["scheduler"] if sched is not None else None)
        return self


class ModelCheckpoint:
    """``save_last`` / ``{epoch}.ckpt`` files in ``dirpath`` (the subset of pl.callbacks.ModelCheckpoint that
    reference train.py:84-89 configures; top-k selection on val_loss needs the private validation data)."""

    def __init__(self, monitor=None, save_last=True, save_top_k=1, dirpath="checkpoints", filename="{epoch}"):
        self.monitor, self.save_last, self.save_top_k = monitor, save_last, save_top_k
        self.dirpath, self.filename = str(dirpath), filename

    def state_dict(self):
        return {"best_k": list(self.__dict__.get("best_k", [])), "best_model_path": getattr(self, "best_model_path", ""),
                "best_model_score": getattr(self, "best_model_score", None)}

    def load_state_dict(self, st):
        self.__dict__["best_k"] = [tuple(b) for b in st.get("best_k", [])]
        if st.get("best_model_path"):
            self.best_model_path, self.best_model_score = st["best_model_path"], st.get("best_model_score")

    def save(self, trainer, model, optimizer, scheduler=None):
        """``last.ckpt`` every epoch; ``epoch=N.ckpt`` kept for the ``save_top_k`` best values of ``monitor`` (min mode, as
        for val_loss; every epoch counts as best when the monitored metric was not logged)."""
        os.makedirs(self.dirpath, exist_ok=True)
        if self.save_top_k != 0:
            score = trainer.logged_metrics.get(self.monitor) if self.monitor else None
            score = float(score) if score is not None else float(-trainer.current_epoch)     # no metric: newest wins
            path = os.path.join(self.dirpath, self.filename.format(epoch=f"epoch={trainer.current_epoch}") + ".ckpt")
            best = self.__dict__.setdefault("best_k", [])
            if self.save_top_k < 0 or len(best) < self.save_top_k or score < max(b[0] for b in best):
                best.append((score, path))
                best.sort(key=lambda t: t[0])
                while 0 < self.save_top_k < len(best):
                    _s, worst = best.pop()
                    if os.path.exists(worst) and worst != path:
                        os.remove(worst)
                self.best_model_path, self.best_model_score = best[0][1], best[0][0]
                if any(b[1] == path for b in best):      # (the checkpoint records this callback's state: build it afterwards)
                    torch.save(model.checkpoint_dict(trainer, optimizer, scheduler), path)
        if self.save_last:
            torch.save(model.checkpoint_dict(trainer, optimizer, scheduler), os.path.join(self.dirpath, "last.ckpt"))
