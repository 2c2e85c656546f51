from . import codec
from .http import HTTPClient, HTTPServer
from .innode import InNodeClient, InNodeServer

__all__ = ["HTTPClient", "HTTPServer", "InNodeClient", "InNodeServer", "codec"]
