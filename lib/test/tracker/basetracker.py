"""Reference lib/test/tracker/basetracker.py:4-21 (the visdom hooks are out of scope)."""


class BaseTracker:
    def __init__(self, params):
        self.params = params
        self.visdom = None

    def predicts_segmentation_mask(self):
        return False

    def initialize(self, image, info: dict) -> dict:
        raise NotImplementedError

    def track(self, image, info: dict = None) -> dict:
        raise NotImplementedError
