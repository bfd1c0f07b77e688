"""reference model/MF.py surface -> sml_amd.mf (class path model.MF.MFbasemode is the
checkpoint contract of --pre_model)."""
from sml_amd.mf import MFbasemode, MF2  # noqa: F401
