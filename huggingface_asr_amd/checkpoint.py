"""Checkpoint averaging in the reference's state-dict names.

Reference behaviour restated (`src/utilities/model_utils.py:54-65`, `src/utilities/general_utils.py:88-101`):
  * the checkpoints are `<experiment_dir>/checkpoint*/pytorch_model.bin`, taken in `glob` order;
  * tensors are summed key by key IN THAT ORDER, accumulating into the first checkpoint's tensor (so the sum carries
    the first checkpoint's dtype), and the sum is divided by the NUMBER OF CHECKPOINTS (also for a key that some
    checkpoints lack; an integer tensor such as a step counter becomes a float by the division);
  * the first checkpoint's directory is copied to `<experiment_dir>/average_checkpoint`, the experiment's `tokenizer/` and
    `feature_extractor/` directories are copied on top (a missing one raises, as `shutil.copytree` does there), and the
    averaged weights are written as `pytorch_model.bin`.
What is added here: a checkpoint directory may hold `model.safetensors` instead of the pickle (what current `transformers`
writes), and `average_into_trainer` loads an average straight into a trainer's flat parameter store on the device.

Host-side glue: tensors are added with torch in the order above, so the result is bit-identical to the reference's for the
same files (IEEE add / divide); nothing of the per-step path goes through this module.
"""
from __future__ import annotations

import glob
import os
import shutil
from typing import Dict, Tuple

import torch

WEIGHT_FILES = ("pytorch_model.bin", "model.safetensors")


def average_dicts(*dicts) -> Tuple[Dict, int]:
    """Key-wise running sum over `dicts` in argument order and the number of dicts (general_utils.py:88-101).
    Accumulates IN PLACE into the first dict's values, like the reference."""
    total: Dict = {}
    for d in dicts:
        for key, value in d.items():
            if key in total:
                total[key] += value
            else:
                total[key] = value
    return total, len(dicts)


def _load_weights(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=True)


def checkpoint_files(experiment_dir: str) -> list[str]:
    """The weight file of every `checkpoint*` directory, in the reference's `glob` order (model_utils.py:55)."""
    found = glob.glob(f"{experiment_dir}/checkpoint*/pytorch_model.bin")
    if not found:
        found = glob.glob(f"{experiment_dir}/checkpoint*/model.safetensors")
    return found


def average_state_dicts(*state_dicts) -> Dict[str, torch.Tensor]:
    """sum (in order, first dict's dtype) / number of dicts (model_utils.py:57-59)."""
    if not state_dicts:
        raise ValueError("average_state_dicts needs at least one state dict")
    total, n = average_dicts(*state_dicts)
    return {key: value.div(n) for key, value in total.items()}


def average_checkpoints(experiment_dir: str) -> str:
    """Average every checkpoint under `experiment_dir` and write `<experiment_dir>/average_checkpoint`; returns that path
    (model_utils.py:54-65).  An experiment without checkpoints raises IndexError, as the reference's `checkpoints[0]` does."""
    checkpoints = checkpoint_files(experiment_dir)
    if not checkpoints:
        raise IndexError(f"no checkpoint*/{{{','.join(WEIGHT_FILES)}}} under {experiment_dir}")
    average = average_state_dicts(*[_load_weights(p) for p in checkpoints])
    dst_path = os.path.join(experiment_dir, "average_checkpoint")
    shutil.copytree(os.path.dirname(checkpoints[0]), dst_path, dirs_exist_ok=True)
    shutil.copytree(os.path.join(experiment_dir, "tokenizer"), dst_path, dirs_exist_ok=True)
    shutil.copytree(os.path.join(experiment_dir, "feature_extractor"), dst_path, dirs_exist_ok=True)
    stale = os.path.join(dst_path, "model.safetensors")          # copied from the first checkpoint: must not shadow the average
    if os.path.exists(stale):
        os.remove(stale)
    torch.save(average, os.path.join(dst_path, "pytorch_model.bin"))
    return dst_path


def average_into_trainer(trainer, *state_dicts) -> Dict[str, torch.Tensor]:
    """Average reference-named state dicts and load the result into `trainer`'s flat store (fp32 masters + bf16 mirrors);
    returns the averaged dict."""
    average = average_state_dicts(*[{k: v.detach().to("cpu").clone() for k, v in sd.items() if torch.is_tensor(v)} for sd in state_dicts])
    trainer.load_state_dict(average)
    return average
