"""Oracle (TEST INFRASTRUCTURE, not product): the quality head of ReLaX-VQA at inference.

Follows /root/reference/src:
  demo_test.py:177-181     imputer.transform (SimpleImputer(mean): NaN -> column mean), scaler.transform
                           (MinMaxScaler: x * scale_ + min_, float64), then torch.tensor(dtype=float32)
  model_regression.py:37-58  Mlp: fc1 -> BatchNorm1d (eval) -> GELU -> [dropout] -> fc2 -> GELU -> [dropout] -> fc3
  demo_test.py:25-35       fix_state_dict: strip 'module.', drop 'n_averaged'
  demo_test.py:211-219     optional rescale (p / 100) * 4 + 1 for youtube_ugc / konvid_1k
Pinned against the reference's own Mlp class and the reference's real KoNViD scaler pickles by oracle/make_golden.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def fix_state_dict(sd):
    out = {}
    for k, v in sd.items():
        if k == "n_averaged":
            continue
        out[k[7:] if k.startswith("module.") else k] = v
    return out


def preprocess(features, imputer_statistics, scale, min_):
    """features [n,F] (any float) -> float32 [n,F] after NaN->mean imputation and min-max scaling in float64."""
    x = np.asarray(features, dtype=np.float64).copy()
    nan = np.isnan(x)
    if imputer_statistics is not None:
        x[nan] = np.broadcast_to(np.asarray(imputer_statistics, dtype=np.float64), x.shape)[nan]
    x = x * np.asarray(scale, dtype=np.float64) + np.asarray(min_, dtype=np.float64)
    return x.astype(np.float32)


@torch.no_grad()
def mlp_forward(sd, x):
    """x float32 [n,F] -> float32 [n] (eval mode: dropout off, BatchNorm running stats)."""
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in fix_state_dict(sd).items()}
    x = torch.as_tensor(x)
    y = F.linear(x, sd["fc1.weight"], sd["fc1.bias"])
    y = F.batch_norm(y, sd["bn1.running_mean"], sd["bn1.running_var"], sd["bn1.weight"], sd["bn1.bias"], False, 0.0, BN_EPS)
    y = F.gelu(y)
    y = F.gelu(F.linear(y, sd["fc2.weight"], sd["fc2.bias"]))
    return F.linear(y, sd["fc3.weight"], sd["fc3.bias"]).squeeze(-1).numpy()


def predict(sd, features, imputer_statistics, scale, min_):
    return mlp_forward(sd, preprocess(features, imputer_statistics, scale, min_))


def rescale_0_100_to_1_5(p):
    return (p / 100.0) * 4 + 1
