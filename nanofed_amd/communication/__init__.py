from . import codec
from .http import ClientEndpoints, HTTPClient, HTTPServer, ServerEndpoints
from .innode import InNodeClient, InNodeServer

__all__ = [
    "HTTPClient",
    "HTTPServer",
    "ClientEndpoints",
    "ServerEndpoints",
    "InNodeClient",
    "InNodeServer",
    "codec",
]
