"""reference evalution/evaluation2.py surface -> sml_amd.evaluation."""
from sml_amd.evaluation import test_model  # noqa: F401
