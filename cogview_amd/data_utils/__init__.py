"""Readers / writers for the token files the trainer consumes: data_utils/datasets.py:63-128,
data_utils/configure_data.py:276-291, and the img2code stage of preprocess/preprocess_text_image_data.py."""
from .datasets import (BinaryDataset, RandomMappingDataset, TextCodeTemplate, get_dataset_by_type,  # noqa: F401
                       write_compact_binary)
from .preprocess import images_to_compact_binary                                                   # noqa: F401
