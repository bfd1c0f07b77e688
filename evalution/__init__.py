"""Drop-in `evalution` package (sic: the reference's spelling) -> sml_amd.evaluation."""
