"""Drop-in alias: `python3 -m annotator {train,evaluate}` runs the MI355X engine (dnncancerannotator_amd)."""
from dnncancerannotator_amd import engine, load, dump, losses, metrics, models, data   # noqa: F401
