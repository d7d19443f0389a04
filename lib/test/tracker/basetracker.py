"""Common base of the trackers: holds the parameter bag and fixes the two-method contract the evaluation code drives
(reference lib/test/tracker/basetracker.py:4-21).  The visdom drawing hooks of the reference are not part of the path."""


class BaseTracker:
    visdom = None

    def __init__(self, params):
        self.params = params

    def predicts_segmentation_mask(self) -> bool:
        return False

    def initialize(self, image, info: dict) -> dict:
        raise NotImplementedError("%s.initialize(image, info)" % type(self).__name__)

    def track(self, image, info: dict = None) -> dict:
        raise NotImplementedError("%s.track(image, info)" % type(self).__name__)
