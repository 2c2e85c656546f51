from .client import ClientEndpoints, HTTPClient
from .server import HTTPServer, ServerEndpoints
from .types import (
    ClientModelUpdateRequest,
    GlobalModelResponse,
    ModelUpdateResponse,
    ServerModelUpdateRequest,
)

__all__ = [
    "HTTPClient",
    "ClientEndpoints",
    "HTTPServer",
    "ServerEndpoints",
    "ClientModelUpdateRequest",
    "ServerModelUpdateRequest",
    "ModelUpdateResponse",
    "GlobalModelResponse",
]
