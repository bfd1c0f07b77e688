"""Drop-in `model` package: the reference's module paths (model.MF, model.conv_transfer,
model.transfer) backed by the MI355X build in sml_amd/."""
