"""Placeholder — filled in below (train_unidefense_model)."""


class AbstractEngine(object):
    path = "engine/abstract_engine.py"
