"""Model registry -- ``get_model_class`` as in the reference
(oct_image_segmentation_models/models/__init__.py:9-22).  Only the U-Net is on the accelerated path;
DeepLabV3+ (ImageNet weights from the network) is out of scope."""
from typing import Type

from . import base_model
from . import unet

model_name_map = {
    unet.UNET_MODEL_NAME: unet.UNet,
}


def get_model_class(model_name: str) -> Type[base_model.BaseModel]:
    if not isinstance(model_name, str):
        raise TypeError("model_name must be a str")
    model_class = model_name_map.get(model_name)
    if model_class is None:
        raise ValueError(f"Model name: '{model_name}' could not be found.")
    return model_class
