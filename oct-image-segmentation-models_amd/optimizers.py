"""Optimizer constructors a caller passes as ``TrainingParams.opt_con`` (the reference receives a Keras
optimizer class there, training/training.py:190-193).  The update itself runs in ``adam_k`` / ``sgd_k``
(Keras formulations)."""
from __future__ import annotations


class Optimizer:
    def get_config(self) -> dict:
        raise NotImplementedError


class Adam(Optimizer):
    def __init__(self, learning_rate: float = 1e-3, beta_1: float = 0.9, beta_2: float = 0.999,
                 epsilon: float = 1e-7, name: str = "Adam", **kwargs):
        if "lr" in kwargs:
            learning_rate = kwargs.pop("lr")
        if kwargs:
            raise TypeError(f"unsupported Adam arguments: {sorted(kwargs)}")
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon, self.name = learning_rate, beta_1, beta_2, epsilon, name

    def get_config(self) -> dict:
        return {"name": self.name, "learning_rate": self.learning_rate, "beta_1": self.beta_1,
                "beta_2": self.beta_2, "epsilon": self.epsilon, "amsgrad": False}

    def apply(self, engine) -> None:
        engine.adam_step(lr=self.learning_rate, beta_1=self.beta_1, beta_2=self.beta_2, epsilon=self.epsilon)


class SGD(Optimizer):
    def __init__(self, learning_rate: float = 1e-2, momentum: float = 0.0, name: str = "SGD", **kwargs):
        if "lr" in kwargs:
            learning_rate = kwargs.pop("lr")
        if kwargs:
            raise TypeError(f"unsupported SGD arguments: {sorted(kwargs)}")
        self.learning_rate, self.momentum, self.name = learning_rate, momentum, name

    def get_config(self) -> dict:
        return {"name": self.name, "learning_rate": self.learning_rate, "momentum": self.momentum, "nesterov": False}

    def apply(self, engine) -> None:
        engine.sgd_step(lr=self.learning_rate, momentum=self.momentum)
