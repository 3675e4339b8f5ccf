class CornacException(Exception):
    pass


class ScoreException(CornacException):
    pass
