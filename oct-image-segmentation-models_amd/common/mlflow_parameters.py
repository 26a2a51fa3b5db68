"""``MLflowParameters`` holder (reference: common/mlflow_parameters.py:4-15).  Experiment tracking is a
network service and out of scope: ``train_model`` accepts ``None`` and ignores a non-None value with a warning."""


class MLflowParameters:
    def __init__(self, tracking_uri: str, username: str = None, password: str = None, experiment: str = None):
        self.tracking_uri, self.username, self.password, self.experiment = tracking_uri, username, password, experiment
