from .dataset import PatientDRRDataset, create_train_val_datasets  # noqa: F401
