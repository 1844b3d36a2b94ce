"""Checkpoint I/O with the contract of detectron2's DetectionCheckpointer (checkpoint/detection_checkpoint.py:17-143 on
top of fvcore's Checkpointer): `.pth` files written by torch.save ({"model": state_dict, "optimizer", "scheduler",
"iteration", ...}) and `.pkl` files in the Detectron2 model-zoo format ({"model": {name: ndarray}, "__author__": ...,
"matching_heuristics": bool}) - the format of U2Seg's `dino_RN50_pretrain_d2_format.pkl` (u2seg_R50_800.yaml:6) and of
its released `cocotrain_*.pth` weights.  The state-dict names of this package are the reference's, so both load 1:1.

Not implemented (raises): Caffe2 / Detectron1 name conversion (`c2_model_loading.convert_c2_detectron_names`), which
the U2Seg configs do not use."""
import logging
import os
import pickle

import numpy as np
import torch

logger = logging.getLogger(__name__)


class _IncompatibleKeys:
    def __init__(self, missing_keys, unexpected_keys, incorrect_shapes):
        self.missing_keys, self.unexpected_keys, self.incorrect_shapes = missing_keys, unexpected_keys, incorrect_shapes


def align_by_suffix(model_keys, ckpt_state):
    """c2_model_loading.align_and_update_state_dicts without the Caffe2 renaming: a checkpoint key is matched to the
    model key it is the longest suffix of (e.g. a backbone-only file 'res2.0.conv1.weight' ->
    'backbone.bottom_up.res2.0.conv1.weight'); exact names win.  Returns a dict keyed by model names."""
    ckpt_keys = sorted(ckpt_state.keys())
    out, used = {}, set()
    for mk in model_keys:
        best = None
        for ck in ckpt_keys:
            if mk == ck or mk.endswith("." + ck):
                if best is None or len(ck) > len(best):
                    best = ck
        if best is not None:
            out[mk] = ckpt_state[best]
            used.add(best)
    for ck in ckpt_keys:  # keep the unmatched ones so that they are reported as unexpected
        if ck not in used:
            out.setdefault(ck, ckpt_state[ck])
    return out


class DetectionCheckpointer:
    def __init__(self, model, save_dir="", *, save_to_disk=True, **checkpointables):
        self.model, self.save_dir, self.save_to_disk = model, save_dir, save_to_disk
        self.checkpointables = dict(checkpointables)  # e.g. optimizer=..., scheduler=...

    # ---- reading ---------------------------------------------------------------------------------------------------
    def _load_file(self, filename):
        if filename.endswith(".pkl"):
            with open(filename, "rb") as f:
                data = pickle.load(f, encoding="latin1")
            if "model" in data and "__author__" in data:
                logger.info("Reading a file from '%s'", data["__author__"])
                return data
            raise NotImplementedError("Caffe2 / Detectron1 .pkl files need name conversion, which U2Seg does not use")
        if filename.endswith(".pyth"):
            raise NotImplementedError("pycls checkpoints are not used by the U2Seg configs")
        loaded = torch.load(filename, map_location="cpu", weights_only=False)
        if "model" not in loaded:
            loaded = {"model": loaded}
        return loaded

    def _load_model(self, checkpoint):
        state = checkpoint.pop("model")
        state = {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in state.items()}
        for k, v in state.items():
            if not isinstance(v, torch.Tensor):
                raise ValueError("Unsupported type found in checkpoint! {}: {}".format(k, type(v)))
        # strip a DataParallel / DDP "module." prefix if every key has it (fvcore Checkpointer._load_model)
        if state and all(k.startswith("module.") for k in state):
            state = {k[len("module."):]: v for k, v in state.items()}
        model_state = self.model.state_dict()
        if checkpoint.get("matching_heuristics", False):
            state = align_by_suffix(list(model_state.keys()), state)
        incorrect = []
        for k in list(state.keys()):
            if k in model_state and tuple(model_state[k].shape) != tuple(state[k].shape):
                incorrect.append((k, tuple(state[k].shape), tuple(model_state[k].shape)))
                state.pop(k)
        res = self.model.load_state_dict(state, strict=False)
        missing = [k for k in res.missing_keys if k not in ("pixel_mean", "pixel_std")]
        unexpected = [k for k in res.unexpected_keys if "anchor_generator.cell_anchors" not in k]
        return _IncompatibleKeys(missing, unexpected, incorrect)

    def load(self, path, checkpointables=None):
        """Loads weights (and, when present and requested, the optimizer / scheduler state); returns the remaining items
        of the checkpoint (e.g. {"iteration": ...}) like the reference."""
        if not path:
            logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        assert os.path.isfile(path), "Checkpoint {} not found!".format(path)
        checkpoint = self._load_file(path)
        incompatible = self._load_model(checkpoint)
        for k in incompatible.incorrect_shapes:
            logger.warning("Skip loading parameter '%s' to the model due to incompatible shapes: %s in the checkpoint but %s "
                           "in the model! You might want to double check if this is expected.", *k)
        if incompatible.missing_keys:
            logger.warning("Some model parameters or buffers are not found in the checkpoint: %s", incompatible.missing_keys)
        if incompatible.unexpected_keys:
            logger.warning("The checkpoint state_dict contains keys that are not used by the model: %s", incompatible.unexpected_keys)
        self.last_incompatible = incompatible
        self._unnest_trainer_state(checkpoint)
        for key in self.checkpointables if checkpointables is None else checkpointables:
            if key in checkpoint:
                self.checkpointables[key].load_state_dict(checkpoint.pop(key))
        opt = self.checkpointables.get("optimizer")
        if opt is not None and hasattr(opt, "refresh_layouts"):
            opt.refresh_layouts()  # the weights changed: rewrite the cached kernel layouts
        return checkpoint

    @staticmethod
    def _unnest_trainer_state(checkpoint):
        """The reference's DefaultTrainer registers ITSELF as the one checkpointable besides the model (engine/defaults.py:389-394):
        its files hold {"model", "trainer": {"iteration", "hooks": {"LRScheduler": <scheduler state>}, "_trainer": {"iteration",
        "optimizer": <torch.optim.SGD state>, ["grad_scaler"]}}, "iteration"} (engine/train_loop.py:195-208,423-430,523-530,
        engine/defaults.py:499-506, engine/hooks.py:365-367), while plain_train_net.py-style files carry "optimizer" /
        "scheduler" at the top level.  Both forms end up as top-level "optimizer" / "scheduler" / "iteration" entries here."""
        tr = checkpoint.get("trainer")
        if not isinstance(tr, dict):
            return
        inner = tr.get("_trainer", {})
        if "optimizer" in inner:
            checkpoint.setdefault("optimizer", inner["optimizer"])
        sched = tr.get("hooks", {}).get("LRScheduler")
        if sched is not None:
            checkpoint.setdefault("scheduler", sched)
        if "iteration" in tr:
            checkpoint.setdefault("iteration", tr["iteration"])

    def has_checkpoint(self):
        """fvcore Checkpointer.has_checkpoint: a `last_checkpoint` file exists in the save directory."""
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def resume_or_load(self, path, *, resume=True):
        last = os.path.join(self.save_dir, "last_checkpoint")
        if resume and os.path.exists(last):
            with open(last) as f:
                path = os.path.join(self.save_dir, f.read().strip())
            return self.load(path)
        return self.load(path, checkpointables=[])

    # ---- writing ---------------------------------------------------------------------------------------------------
    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return None
        data = {"model": self.model.state_dict()}
        for key, obj in self.checkpointables.items():
            data[key] = obj.state_dict()
        data.update(kwargs)
        if "optimizer" in data and "iteration" in data and "trainer" not in data:
            # the same state once more in the nesting a reference DefaultTrainer reads back with --resume (see
            # _unnest_trainer_state; the tensors are shared with the top-level entries, torch.save stores them once)
            tr = {"iteration": data["iteration"], "_trainer": {"iteration": data["iteration"], "optimizer": data["optimizer"]}}
            if "scheduler" in data:
                tr["hooks"] = {"LRScheduler": data["scheduler"]}
            data["trainer"] = tr
        basename = "{}.pth".format(name)
        os.makedirs(self.save_dir, exist_ok=True)
        save_file = os.path.join(self.save_dir, basename)
        torch.save(data, save_file)
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(basename)
        return save_file
