INTER_NEAREST = 0
INTER_LINEAR = 1
INTER_AREA = 3
def __getattr__(name):
    raise AttributeError(f"cv2 stub has no {name}")
