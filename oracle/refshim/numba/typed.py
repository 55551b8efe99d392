"""numba.typed stand-in (see package docstring): a typed List is a plain list."""
List = list
