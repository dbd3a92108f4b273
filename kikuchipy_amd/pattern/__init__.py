from kikuchipy_amd.pattern._pattern import (  # noqa: F401
    remove_dynamic_background,
    remove_static_background,
)
