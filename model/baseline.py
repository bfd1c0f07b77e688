"""reference model/baseline.py: only SPMF.run_one_stage2 (the bare-MF fine-tune / full-retrain step) is hosted -> sml_amd.baseline."""
from sml_amd.baseline import SPMF, offlineDataset_withsample  # noqa: F401
