"""Namespace of the reference's `main` package: only `main.backend` is provided here (integration/README.md)."""
