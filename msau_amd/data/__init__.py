"""Chargrid input pipeline counterparts (CPU, API-compatible with the reference's
funsd_preprocessing_word_level.py / data_generator_funsd_bert.py)."""
from .funsd import (CellNode, FUNSDBertDataLoaderBoxMaskBoxLabel, FUNSDCharGridDataLoaderBoxMaskBoxLabel, FUNSDMaskDataLoader,  # noqa: F401
                    get_box_mask_box_label,
                    get_box_mask_box_label_word, get_charset, get_preprocessed_list_word_msau,
                    transform_from_charset)
