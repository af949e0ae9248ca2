class MidiFormatError(Exception):
    """A MIDI file the model's representation cannot hold (tempo/key/time-signature change mid-song, ...)."""

    def __init__(self, msg=''):
        super().__init__(msg)
        self.msg = msg
