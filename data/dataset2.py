"""reference data/dataset2.py surface -> sml_amd.datasets."""
from sml_amd.datasets import transfer_data, testDataset, trainDataset_withPreSample  # noqa: F401
