from nanofed_amd.core.types import ModelVersion

from .manager import ModelManager, make_json_serializable

__all__ = ["ModelManager", "ModelVersion", "make_json_serializable"]
