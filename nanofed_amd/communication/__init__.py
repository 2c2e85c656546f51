from .http import HTTPClient, HTTPServer

__all__ = ["HTTPClient", "HTTPServer"]
