from .evaluation import evaluate_model, eval_model  # noqa: F401  (``eval_model`` is the README / north_star alias)
