"""How decisively do the dropout-statistics checks reject a wrong configuration?  Prints each criterion's value against
its threshold for the reference settings and for the negative controls of tests/test_train_grads.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_train_grads import _dropout_stats

def report(tag, ov, fixture):
    meta, fx, n, lm, ls, mean, var = _dropout_stats(ov, fixture)
    se = (meta["loss_std"] ** 2 / n + ls ** 2 / n) ** 0.5
    se_el = np.sqrt((fx["out_var"] + var) / n)
    ratio = (var + 1e-6) / (fx["out_var"] + 1e-6)
    print(f"{tag:28s} loss-mean z={abs(lm - meta['loss_mean']) / se:6.2f} (<=4)  loss-std ratio={ls / meta['loss_std']:5.2f} (0.8..1.25)  "
          f"var-mean ratio={float(var.mean()) / meta['out_var_mean']:5.2f} (0.9..1.1)  worst elem z={np.max((np.abs(mean - fx['out_mean']) - 1e-3) / se_el):6.2f} (<=5)  "
          f"var ratio min/max={ratio.min():4.2f}/{ratio.max():4.2f} (0.6..1.6)", flush=True)

f1, f2 = "g13_dropout_stats.npz", "g13_embed_goal_drop_stats.npz"
report("reference settings", {}, f1)
for ov in (dict(attn_pdrop=0.0), dict(resid_pdrop=0.0), dict(mlp_pdrop=0.3), dict(mlp_pdrop=0.6)):
    report(str(ov), ov, f1)
report("embed/goal reference", {}, f2)
for ov in (dict(embed_pdrob=0.0), dict(goal_drop=0.0), dict(goal_drop=0.6), dict(embed_pdrob=0.4)):
    report(str(ov), ov, f2)
