"""reference model/conv_transfer.py surface -> sml_amd.conv_transfer."""
from sml_amd.conv_transfer import (Gelu, one_transfer, ConvTransfer_com, ConvTransfer,  # noqa: F401
                                   ConvTransfer_com2, ConvTransfer_com3, one_transfer_com)
