from .cornell import cornell_box  # noqa: F401
from .bathroom import generate_room_scene  # noqa: F401
