"""Checkpoint averaging: `huggingface_asr_amd.checkpoint` against the behaviour of the reference's
`average_checkpoints` / `average_dicts` (model_utils.py:54-65, general_utils.py:88-101), restated here in numpy: sum in glob
order in the first checkpoint's dtype, divide by the number of checkpoints, copy the first checkpoint + tokenizer +
feature_extractor directories and write `pytorch_model.bin`."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import seeded_state_dict
from huggingface_asr_amd import checkpoint as C
from huggingface_asr_amd import shapes


def _experiment(tmp_path, n, fmt="bin", with_aux=True):
    cfg = dict(shapes.TINY)
    sds = []
    for i in range(n):
        sd = seeded_state_dict(cfg, 100 + i)
        sd["step_counter"] = torch.tensor(10 * (i + 1), dtype=torch.int64)      # an integer entry: becomes a float by the division
        d = tmp_path / f"checkpoint-{(i + 1) * 500}"
        d.mkdir()
        (d / "config.json").write_text('{"ckpt": %d}' % i)
        if fmt == "bin":
            torch.save(sd, d / "pytorch_model.bin")
        else:
            from safetensors.torch import save_file
            save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
        sds.append(sd)
    if with_aux:
        (tmp_path / "tokenizer").mkdir()
        (tmp_path / "tokenizer" / "tokenizer.json").write_text("{}")
        (tmp_path / "feature_extractor").mkdir()
        (tmp_path / "feature_extractor" / "preprocessor_config.json").write_text("{}")
    return sds


def _expected(sds_in_order):
    out = {}
    for k in sds_in_order[0]:
        acc = sds_in_order[0][k].numpy().copy()
        for sd in sds_in_order[1:]:
            acc = acc + sd[k].numpy().astype(acc.dtype)
        out[k] = acc
    return out, len(sds_in_order)


@pytest.mark.parametrize("fmt", ["bin", "safetensors"])
def test_average_checkpoints_matches_the_reference_recipe(tmp_path, fmt):
    sds = _experiment(tmp_path, 3, fmt)
    dst = C.average_checkpoints(str(tmp_path))
    assert dst == os.path.join(str(tmp_path), "average_checkpoint")
    # order = the glob order the reference sums in
    pat = "pytorch_model.bin" if fmt == "bin" else "model.safetensors"
    order = [int(os.path.basename(os.path.dirname(p)).split("-")[1]) // 500 - 1 for p in glob.glob(f"{tmp_path}/checkpoint*/{pat}")]
    tot, n = _expected([sds[i] for i in order])
    got = torch.load(os.path.join(dst, "pytorch_model.bin"), weights_only=True)
    assert set(got) == set(tot)
    for k, v in tot.items():
        if k == "step_counter":
            assert got[k].is_floating_point() and float(got[k]) == pytest.approx(20.0)
            continue
        want = torch.from_numpy(v).div(n)
        assert got[k].dtype == torch.float32 and torch.equal(got[k], want), k               # bit-exact: same adds, same divide
    # the directory is the first checkpoint's plus tokenizer / feature extractor files, and no stale weight file
    assert os.path.exists(os.path.join(dst, "config.json"))
    assert os.path.exists(os.path.join(dst, "tokenizer.json")) and os.path.exists(os.path.join(dst, "preprocessor_config.json"))
    assert not os.path.exists(os.path.join(dst, "model.safetensors"))


def test_average_dicts_accumulates_in_place_and_counts_dicts():
    a = {"w": torch.tensor([1.0, 2.0]), "only_a": torch.tensor([4.0])}
    b = {"w": torch.tensor([3.0, 6.0])}
    wa = a["w"]
    tot, n = C.average_dicts(a, b)
    assert n == 2 and tot["w"] is wa and torch.equal(wa, torch.tensor([4.0, 8.0]))          # in place, like the reference
    avg = C.average_state_dicts({"w": torch.tensor([1.0, 2.0]), "only_a": torch.tensor([4.0])}, b)
    assert torch.equal(avg["w"], torch.tensor([2.0, 4.0]))
    assert torch.equal(avg["only_a"], torch.tensor([2.0]))                                   # divided by the number of dicts, not of holders


def test_error_behaviour(tmp_path):
    with pytest.raises(IndexError):
        C.average_checkpoints(str(tmp_path))                                                 # no checkpoints: `checkpoints[0]` in the reference
    _experiment(tmp_path, 2, with_aux=False)
    with pytest.raises(FileNotFoundError):
        C.average_checkpoints(str(tmp_path))                                                 # no tokenizer directory: copytree raises there too
    with pytest.raises(ValueError):
        C.average_state_dicts()


def test_against_the_reference_average_checkpoints_fixture(tmp_path):
    """tests/golden/ckpt_avg.npz was written by the REFERENCE's `average_checkpoints` (make_golden.py `run_ckpt_average_case`: model_utils.py:54-65 +
    general_utils.py:88-101 run in a temporary experiment directory).  Same checkpoints rebuilt from the seeds, summed in the order the reference's glob
    returned: bit-identical tensors, dtypes (bf16 stays bf16, the int64 counter becomes float32), a key one checkpoint lacks still divided by 3."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from helpers import load_golden
    from huggingface_asr_amd import synth
    g = load_golden("ckpt_avg")

    def sds():
        out = {}
        for i in range(3):
            sd = {"enc.weight": torch.from_numpy(synth.normal(40 + i, "ck_w", (7, 5), 1.0)), "enc.bias": torch.from_numpy(synth.normal(40 + i, "ck_b", (5,), 0.3)),
                  "head.weight": torch.from_numpy(synth.normal(40 + i, "ck_h", (4, 6), 2.0)).to(torch.bfloat16), "step_counter": torch.tensor(10 * (i + 1) + i, dtype=torch.int64)}
            if i != 1:
                sd["extra.scale"] = torch.from_numpy(synth.normal(40 + i, "ck_e", (3,), 1.0))
            out[f"checkpoint-{(i + 1) * 500}"] = sd
        return out
    by_name = sds()
    avg = C.average_state_dicts(*[by_name[str(n)] for n in g["glob_order"]])
    keys = sorted(k[4:] for k in g.files if k.startswith("avg/"))
    assert sorted(avg) == keys
    for k in keys:
        assert str(avg[k].dtype) == str(g["dtype/" + k]), k
        got = avg[k].float().numpy() if avg[k].dtype == torch.bfloat16 else avg[k].numpy()
        np.testing.assert_array_equal(got, g["avg/" + k], err_msg=k)
    # the directory-level entry point: same files in the output, first checkpoint's config, same numbers (this file system's glob order may differ: allclose)
    for name, sd in sds().items():
        (tmp_path / name).mkdir()
        torch.save(sd, tmp_path / name / "pytorch_model.bin")
        (tmp_path / name / "config.json").write_text('{"ckpt": %d}' % (int(name.split("-")[1]) // 500 - 1))
    for aux, fn in (("tokenizer", "tokenizer.json"), ("feature_extractor", "preprocessor_config.json")):
        (tmp_path / aux).mkdir()
        (tmp_path / aux / fn).write_text("{}")
    dst = C.average_checkpoints(str(tmp_path))
    assert sorted(os.listdir(dst)) == list(g["files"])
    first = os.path.basename(os.path.dirname(C.checkpoint_files(str(tmp_path))[0]))
    assert open(os.path.join(dst, "config.json")).read() == '{"ckpt": %d}' % (int(first.split("-")[1]) // 500 - 1)
    out = torch.load(os.path.join(dst, "pytorch_model.bin"), weights_only=True)
    for k in keys:
        np.testing.assert_allclose(out[k].float().numpy(), g["avg/" + k].astype(np.float32), rtol=1e-6, atol=1e-7 if out[k].dtype != torch.bfloat16 else 2e-2)
