"""reference model/transfer.py surface -> sml_amd.driver (class meta_train)."""
from sml_amd.driver import meta_train, SampleDaset, PreSampleDatast  # noqa: F401
from sml_amd.evaluation import test_model  # noqa: F401
from sml_amd.conv_transfer import ConvTransfer, ConvTransfer_com, ConvTransfer_com2  # noqa: F401
from sml_amd.datasets import transfer_data, testDataset  # noqa: F401
