"""Host-side rules of the reference-named API that need no GPU: frame-id parsing, missing weights, empty shards."""
import pytest

import relax_vqa_amd  # noqa: F401


def test_frame_number_follows_the_references_split_rules():
    """src/extractor/visualise_resnet.py:63-79 splits the FILE NAME at '_' and strips the extension from the last part only."""
    from relax_vqa_amd.extractor.visualise_resnet import _frame_number
    assert _frame_number("video_12.png") == 12
    assert _frame_number("5636101558_3_next.png") == "3_next"
    assert _frame_number("5636101558_3_residual.png") == "3_residual"
    assert _frame_number("5636101558_3_residual_of.png") == "3_residual_of"
    assert _frame_number("5636101558_3_residual_imp.png") == "3_residual_imp"
    assert _frame_number("5636101558_3_residual_of_imp.png") == "3_residual_of_imp"
    assert _frame_number("5636101558_3_residual_merged_frag.png") == "3_residual_merged_frag"
    assert _frame_number("5636101558_3_ori_frag.png") == "3_ori_frag"
    # a dot earlier in the base name survives (the reference only strips what follows the first dot of the LAST part)
    assert _frame_number("clip.v2_7_next.png") == "7_next"
    assert _frame_number("clip.v2_7.png") == 7


def test_missing_weight_files_are_an_error_unless_synthetic_weights_are_allowed(monkeypatch):
    from relax_vqa_amd import runtime
    monkeypatch.delenv("RELAX_RESNET50_WEIGHTS", raising=False)
    monkeypatch.delenv("RELAX_ALLOW_SYNTHETIC_WEIGHTS", raising=False)
    with pytest.raises(RuntimeError, match="RELAX_RESNET50_WEIGHTS"):
        runtime._weights_or_synthetic("RELAX_RESNET50_WEIGHTS", "ResNet-50", dict)
    monkeypatch.setenv("RELAX_ALLOW_SYNTHETIC_WEIGHTS", "1")
    assert runtime._weights_or_synthetic("RELAX_RESNET50_WEIGHTS", "ResNet-50", lambda: {"ok": 1}) == {"ok": 1}


def test_fewer_clips_than_ranks_raises_on_every_rank_before_any_work():
    from relax_vqa_amd import distributed as rd
    calls = []
    for rank in range(4):
        with pytest.raises(ValueError, match="n_clips=2 < world=4"):
            rd.extract_dataset(lambda i: calls.append(i), 2, rank, 4)
    assert calls == []          # not even the ranks that own a clip start extracting


def test_process_wide_engine_follows_local_rank(monkeypatch):
    """runtime.get_engine() sits on GPU LOCAL_RANK (one process per GPU); checked through the device argument it builds."""
    from relax_vqa_amd import runtime
    seen = {}

    class Fake:
        def __init__(self, device):
            seen["device"] = device

    import relax_vqa_amd.engine as eng_mod
    monkeypatch.setattr(eng_mod, "RelaxEngine", Fake)
    monkeypatch.setitem(runtime._state, "engine", None)
    monkeypatch.setenv("LOCAL_RANK", "5")
    runtime.get_engine()
    assert seen["device"] == 5
    monkeypatch.setitem(runtime._state, "engine", None)


def test_checkpoint_file_forms_unwrap_to_the_same_state_dict(tmp_path):
    """runtime._load_file: plain state dict, {"state_dict": ...}, DINO's {"teacher": {"backbone.*"}}, DataParallel's module.* -
    all give the plain torchvision / DINO keys (src/extractor/visualise_resnet.py:21, visualise_vit_layer.py:326-328)."""
    import numpy as np
    import torch
    from relax_vqa_amd import runtime
    sd = {"conv1.weight": torch.randn(4, 3, 2, 2), "bn1.running_var": torch.rand(4), "bn1.num_batches_tracked": torch.tensor(3)}
    forms = {
        "plain.pth": sd,
        "wrapped.pth": {"state_dict": sd, "epoch": 12},
        "module.pth": {"module." + k: v for k, v in sd.items()},
        "dino.pth": {"teacher": {"backbone." + k: v for k, v in sd.items()}, "student": {"module.backbone.x": torch.zeros(1)}},
        "both.pth": {"model": {"module.backbone." + k: v for k, v in sd.items()}},
    }
    for name, obj in forms.items():
        path = str(tmp_path / name)
        torch.save(obj, path)
        got = runtime._load_file(path)
        assert set(got) == set(sd), name
        for k in sd:
            assert np.array_equal(got[k], sd[k].numpy()), (name, k)
    torch.save([1, 2, 3], str(tmp_path / "list.pth"))
    with pytest.raises(RuntimeError, match="expected a state dict"):
        runtime._load_file(str(tmp_path / "list.pth"))
