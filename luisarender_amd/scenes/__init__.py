from .cornell import cornell_box  # noqa: F401
from .bathroom import generate_room_scene  # noqa: F401
from .configs import generate_bedroom_scene, generate_camera_scene, generate_kitchen_scene  # noqa: F401
