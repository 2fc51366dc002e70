"""The PRODUCT's metrics (bayesrul_amd/results/metrics.py, called inside every train / val / test step) and the
host-side `aggregate_predictions`, against fixtures generated from the reference's own functions
(tests/golden/make_golden.py: ref_metrics.npz, ref_rmsce.npz) and against the oracle's restatements."""
import os

import numpy as np
import torch

from bayesrul_amd.results import metrics as M
from oracle import restatement as R


def test_sharpness_and_nasa_score_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_metrics.npz"))
    s, yt, yp = (torch.from_numpy(z[k]) for k in ("s", "yt", "yp"))
    np.testing.assert_allclose(M.sharpness(s).numpy(), z["sharp"], rtol=1e-6)
    np.testing.assert_allclose(M.nasa_score(yt, yp).numpy(), z["nasa"], rtol=1e-6)


def test_rms_calibration_error_matches_reference(golden_dir):
    """results/metrics.py:216-274 run on CPU through the get_device shim of make_golden.py."""
    z = np.load(os.path.join(golden_dir, "ref_rmsce.npz"))
    std, yt, yp = (torch.from_numpy(z[k]) for k in ("std", "yt", "yp"))
    for tag, f in (("cal", 1.0), ("over", 0.3), ("under", 3.0)):
        np.testing.assert_allclose(M.rms_calibration_error(yp, std * f, yt).numpy(), z["rmsce_" + tag], rtol=1e-6)
        np.testing.assert_allclose(R.rms_calibration_error(yp, std * f, yt).numpy(), z["rmsce_" + tag], rtol=1e-6)
    # the quantile variant is not pinned by a fixture: product == a direct restatement
    exp_p, obs_p = M.get_proportion_lists(yp, std, yt, 50, "quantile")
    nrm = torch.distributions.Normal(0.0, 1.0)
    ref = ((yp - yt) / std).reshape(-1, 1) <= nrm.icdf(torch.linspace(0, 1, 50))
    assert torch.equal(obs_p, ref.sum(0) / yp.numel())


def test_aggregate_predictions_matches_oracle():
    """HeteroskedasticGaussian.aggregate_predictions (U4, parity unpinned: TyXe absent): the product's host
    function equals the oracle's restatement."""
    from bayesrul_amd.models.bayesian import aggregate_predictions
    g = torch.Generator().manual_seed(3)
    preds = torch.rand(7, 33, 2, generator=g, dtype=torch.float64) + 0.1
    a, b = aggregate_predictions(preds), R.aggregate_predictions(preds)
    assert torch.allclose(a, b, rtol=1e-12)


def test_deep_ensemble_matches_reference(golden_dir):
    """bayesrul/models/deepens.py:9-49 (ref_deepens.npz, generated from the reference's own functions): the moment-matched
    mixture of one method's models, and the ensembles `deep_ensemble_gen` draws (random.seed(1), same combinations)."""
    import pandas as pd

    from bayesrul_amd.models.deepens import deep_ensemble, deep_ensemble_gen
    z = np.load(os.path.join(golden_dir, "ref_deepens.npz"))
    n, rows, k0 = len(z["labels"]), [], 0
    for method in ("HNN", "MCD"):
        for k in range(5):
            rows.append(pd.DataFrame({"method": method, "model": f"{method}_{k:03d}", "labels": z["labels"],
                                      "preds": z["preds"][k0:k0 + n], "stds": z["stds"][k0:k0 + n]}))
            k0 += n
    df = pd.concat(rows, ignore_index=True)
    one = deep_ensemble(df.query("method=='HNN'"))
    np.testing.assert_allclose(one.preds.values, z["one_preds"], rtol=1e-12)
    np.testing.assert_allclose(one.stds.values, z["one_stds"], rtol=1e-12)
    gens = list(deep_ensemble_gen(df, ["HNN", "MCD"], 3, 4))
    assert [g.model.iloc[0] for g in gens] == list(z["gen_models"]) and all((g.method == "DE").all() for g in gens)
    np.testing.assert_allclose(np.stack([g.preds.values for g in gens]), z["gen_preds"], rtol=1e-12)
    np.testing.assert_allclose(np.stack([g.stds.values for g in gens]), z["gen_stds"], rtol=1e-12)
