from .training import train_model, train  # noqa: F401  (``train`` is the alias BASELINE.json's north_star names)
