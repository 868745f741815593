from .cornell import cornell_box  # noqa: F401
