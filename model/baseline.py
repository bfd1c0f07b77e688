"""reference model/baseline.py surface (full-retrain / fine-tune / SPMF MF baselines) -> sml_amd.baseline."""
from sml_amd.baseline import (SPMF, Reservious, StreamingData, get_parse, main, offlineDataset_withsample,  # noqa: F401
                              test_hit_new)

if __name__ == "__main__":
    main()
