"""``UNet`` model container -- constructor, config and preprocess contract of the reference's
``oct_image_segmentation_models/models/unet.py:61-153``; ``build_model()`` returns the HIP-engine-backed
``Model`` instead of a ``tf.keras.Model``."""
from __future__ import annotations

from typing import Callable, Union

from .base_model import BaseModel

UNET_MODEL_NAME = "unet"


class UNet(BaseModel):
    def __init__(self, *, input_channels: int, num_classes: int, image_height: int, image_width: int,
                 start_neurons: int = 8, pool_layers: int = 4, conv_layers: int = 2,
                 enc_kernel: Union[list, tuple] = (3, 3), dec_kernel: Union[list, tuple] = (2, 2)) -> None:
        super().__init__(input_channels=input_channels, num_classes=num_classes,
                         image_height=image_height, image_width=image_width)
        for name, v in (("start_neurons", start_neurons), ("pool_layers", pool_layers), ("conv_layers", conv_layers)):
            if not isinstance(v, int) or isinstance(v, bool):
                raise TypeError(f"{name} must be an int")   # the reference enforces types with @typechecked
        self.start_neurons = start_neurons
        self.pool_layers = pool_layers
        self.conv_layers = conv_layers
        self.enc_kernel = tuple(enc_kernel)
        self.dec_kernel = tuple(dec_kernel)

    def get_preprocess_input_fn(self) -> Callable:
        def preprocess_input_inner(x):
            return x / 255.0

        return preprocess_input_inner

    def get_config(self) -> dict:
        config = super().get_config()
        config.update({
            "start_neurons": self.start_neurons,
            "pool_layers": self.pool_layers,
            "conv_layers": self.conv_layers,
            "enc_kernel": self.enc_kernel,
            "dec_kernel": self.dec_kernel,
        })
        return config

    def build_model(self):
        from .engine_model import Model
        return Model(name=UNET_MODEL_NAME, config=self.get_config())
