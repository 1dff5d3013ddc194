"""blur_jax/utils.py: get_data_shape (the same helper as the CLD package's, cld_jax/utils.py:172-177)."""
from ..cld.utils import get_data_shape  # noqa: F401
