"""utils.* -- host-side helpers with the reference's names (utils/functional.py, dataprep.py,
metrics.py, metrics2.py, config.py)."""
