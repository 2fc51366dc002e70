"""Deep-ensemble aggregation — drop-ins for `bayesrul.models.deepens.deep_ensemble / deep_ensemble_gen`
(bayesrul/models/deepens.py:9-49): the test outputs of M base learners (one row per test window and model, columns
`model, method, labels, preds, stds`) are merged into one Gaussian-mixture moment match per window,
    mu = mean_m mu_m,   sigma^2 = mean_m (mu_m^2 + sigma_m^2) - mu^2.
Host functions (pandas in, pandas out), as in the reference: this is post-processing of prediction tables, not part of
the device path."""
import random
from itertools import combinations
from typing import Iterator, List

import numpy as np
import pandas as pd


def deep_ensemble(df: pd.DataFrame) -> pd.DataFrame:
    """deepens.py:9-30.  `df`: rows of the base learners' test outputs; every model lists the windows in the same order."""
    labels, mus, sigmas = None, [], []
    for _, rows in df.groupby("model"):
        if labels is None:
            labels = rows.labels.values
        mus.append(rows.preds.values)
        sigmas.append(rows.stds.values)
    mu_m, sigma_m = np.stack(mus), np.stack(sigmas)
    mu = mu_m.mean(axis=0)
    sigma = np.sqrt((mu_m**2 + sigma_m**2).mean(axis=0) - mu**2)
    return pd.DataFrame({"preds": mu, "labels": labels, "stds": sigma})


def deep_ensemble_gen(df: pd.DataFrame, base_learners: List[str], n_models_per_ens: int,
                      max_deepens: int) -> Iterator[pd.DataFrame]:
    """deepens.py:33-49: for every method, `max_deepens` ensembles of `n_models_per_ens` of its models `<method>_<k:03d>`,
    drawn with `random.seed(1)` from the combinations in lexicographic order (the reference's draw, reproduced)."""
    random.seed(1)
    for method in base_learners:
        n = len(df.query(f"method=='{method}'").groupby("model"))
        comb = list(combinations(range(n), n_models_per_ens))
        for i, ens in enumerate(random.sample(comb, max_deepens)):
            models = [f"{method}_{k:03d}" for k in ens]
            yield deep_ensemble(df.query(f"model in {models}")).assign(method="DE", model=f"DE_{i:03d}")
