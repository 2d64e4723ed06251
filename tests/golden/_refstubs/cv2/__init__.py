"""Stand-in so that the reference's utils.py (`import cv2`) imports in the build container; never called."""


def resize(*a, **k):
    raise NotImplementedError
